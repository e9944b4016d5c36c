// rpf_collective.cpp -- the scan reducer of include/rpf_engine.h: north_star's "final RCCL reduce over
// xGMI of the per-bin power accumulators" for the single-process, one-engine-per-device form of a scan
// (ncclCommInitAll, ONE ncclReduce(sum, ncclDouble, hops x N) onto the first device, one
// device-to-host copy).  RCCL is loaded with dlopen: a host without librccl.so, or a device list RCCL
// refuses (the same device twice), makes rpf_scan_reducer_create fail with RPF_ERR_HARDWARE and the
// caller keeps adding the per-device spectra on the host (rpf_power --gpus does).
//
// What each device contributes is a block of hops x N doubles: zeroed at the start of a pass
// (rpf_scan_reducer_begin), row h overwritten with an engine's accumulator after that engine finished
// its share of hop h (rpf_scan_reducer_deposit, device-to-device on the engine's device), rows of hops
// another device owns left at zero -- so the sum over devices is the scan.  The sum is taken in RCCL's
// ring order, fixed for a given device list (reproducible run to run; different from the host's device
// order by rounding, ~1e-16).
#include "../../include/rpf_engine.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <string>
#include <vector>

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string* why)
    {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            *why = std::string("librccl.so not found (") + dlerror() + ")";
            return false;
        }
#define RPF_SYM(field, sym)                                           \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));       \
    if (!field) {                                                     \
        *why = std::string("librccl.so lacks ") + sym;                \
        return false;                                                 \
    }
        RPF_SYM(CommInitAll, "ncclCommInitAll")
        RPF_SYM(CommDestroy, "ncclCommDestroy")
        RPF_SYM(Reduce, "ncclReduce")
        RPF_SYM(GroupStart, "ncclGroupStart")
        RPF_SYM(GroupEnd, "ncclGroupEnd")
        RPF_SYM(GetErrorString, "ncclGetErrorString")
#undef RPF_SYM
        return true;
    }
};

thread_local std::string g_reducer_error;

}  // namespace

struct rpf_scan_reducer {
    Rccl rccl;
    int N = 0, max_hops = 0;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<double*> blocks;     // [device slot]: max_hops x N doubles on that device
    double* d_result = nullptr;      // on devices[0]: the reduced block
    std::string last_error;
};

namespace {

int rfail(rpf_scan_reducer* r, int rc, const std::string& msg)
{
    if (r) r->last_error = msg;
    g_reducer_error = msg;
    return rc;
}

void release(rpf_scan_reducer* r)
{
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < r->devices.size(); ++i) {
        if (hipSetDevice(r->devices[i]) != hipSuccess) continue;
        if (i < r->blocks.size() && r->blocks[i]) (void)hipFree(r->blocks[i]);
        if (i < r->streams.size() && r->streams[i]) (void)hipStreamDestroy(r->streams[i]);
        if (i == 0 && r->d_result) (void)hipFree(r->d_result);
    }
    for (ncclComm_t c : r->comms)
        if (c && r->rccl.CommDestroy) (void)r->rccl.CommDestroy(c);
    if (r->rccl.lib) dlclose(r->rccl.lib);
    if (prev >= 0) (void)hipSetDevice(prev);
}

}  // namespace

extern "C" {

const char* rpf_scan_reducer_last_error(const rpf_scan_reducer* r)
{
    return r ? r->last_error.c_str() : g_reducer_error.c_str();
}

int rpf_scan_reducer_create(const int* devices, int n_devices, int N, int max_hops, rpf_scan_reducer** out)
{
    if (!out) return rfail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_scan_reducer_create: out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || N < 2 || max_hops < 1)
        return rfail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_scan_reducer_create: bad argument");
    rpf_scan_reducer* r = new rpf_scan_reducer();
    r->N = N;
    r->max_hops = max_hops;
    r->devices.assign(devices, devices + n_devices);
    std::string why;
    int prev = -1;
    (void)hipGetDevice(&prev);
    auto bail = [&](const std::string& msg) {
        release(r);
        delete r;
        if (prev >= 0) (void)hipSetDevice(prev);
        return rfail(nullptr, RPF_ERR_HARDWARE, msg);
    };
    if (!r->rccl.load(&why)) return bail("RCCL: " + why);
    r->comms.assign(n_devices, nullptr);
    ncclResult_t nrc = r->rccl.CommInitAll(r->comms.data(), n_devices, devices);
    if (nrc != ncclSuccess) {
        for (auto& c : r->comms) c = nullptr;          // (nothing usable to destroy)
        return bail(std::string("ncclCommInitAll: ") + r->rccl.GetErrorString(nrc));
    }
    r->streams.assign(n_devices, nullptr);
    r->blocks.assign(n_devices, nullptr);
    const size_t bytes = sizeof(double) * static_cast<size_t>(N) * max_hops;
    for (int i = 0; i < n_devices; ++i) {
        hipError_t err = hipSetDevice(devices[i]);
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&r->streams[i], hipStreamNonBlocking);
        if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&r->blocks[i]), bytes);
        if (err == hipSuccess && i == 0) err = hipMalloc(reinterpret_cast<void**>(&r->d_result), bytes);
        if (err != hipSuccess) return bail(std::string("scan reducer, device ") + std::to_string(devices[i]) + ": " + hipGetErrorString(err));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    *out = r;
    return RPF_OK;
}

void rpf_scan_reducer_destroy(rpf_scan_reducer* r)
{
    if (!r) return;
    release(r);
    delete r;
}

int rpf_scan_reducer_begin(rpf_scan_reducer* r)
{
    if (!r) return rfail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_scan_reducer_begin: NULL");
    int prev = -1;
    (void)hipGetDevice(&prev);
    const size_t bytes = sizeof(double) * static_cast<size_t>(r->N) * r->max_hops;
    int rc = RPF_OK;
    for (size_t i = 0; i < r->devices.size() && rc == RPF_OK; ++i) {
        hipError_t err = hipSetDevice(r->devices[i]);
        if (err == hipSuccess) err = hipMemsetAsync(r->blocks[i], 0, bytes, r->streams[i]);
        if (err == hipSuccess) err = hipStreamSynchronize(r->streams[i]);
        if (err != hipSuccess) rc = rfail(r, RPF_ERR_HARDWARE, std::string("rpf_scan_reducer_begin: ") + hipGetErrorString(err));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

int rpf_scan_reducer_deposit(rpf_scan_reducer* r, int slot, int hop, const rpf_engine* e)
{
    if (!r || !e || slot < 0 || slot >= static_cast<int>(r->devices.size()) || hop < 0 || hop >= r->max_hops)
        return rfail(r, RPF_ERR_INVALID_ARGUMENT, "rpf_scan_reducer_deposit: bad argument");
    // (the engine's accumulator lives on the slot's device; copied with the plain device-to-device entry below)
    return rpf_copy_power_device(e, r->blocks[slot] + static_cast<size_t>(hop) * r->N, r->streams[slot], r->devices[slot]);
}

int rpf_scan_reducer_reduce(rpf_scan_reducer* r, int hops, double* host_out)
{
    if (!r || !host_out || hops < 1 || hops > r->max_hops)
        return rfail(r, RPF_ERR_INVALID_ARGUMENT, "rpf_scan_reducer_reduce: bad argument");
    int prev = -1;
    (void)hipGetDevice(&prev);
    const size_t count = static_cast<size_t>(hops) * r->N;
    int rc = RPF_OK;
    ncclResult_t nrc = r->rccl.GroupStart();
    for (size_t i = 0; i < r->devices.size() && nrc == ncclSuccess; ++i) {
        if (hipSetDevice(r->devices[i]) != hipSuccess) {
            rc = rfail(r, RPF_ERR_HARDWARE, "rpf_scan_reducer_reduce: hipSetDevice");
            break;
        }
        nrc = r->rccl.Reduce(r->blocks[i], i == 0 ? r->d_result : r->blocks[i], count, ncclDouble, ncclSum, 0, r->comms[i],
                             r->streams[i]);
    }
    const ncclResult_t end = r->rccl.GroupEnd();
    if (nrc == ncclSuccess) nrc = end;
    if (rc == RPF_OK && nrc != ncclSuccess) rc = rfail(r, RPF_ERR_HARDWARE, std::string("ncclReduce: ") + r->rccl.GetErrorString(nrc));
    for (size_t i = 0; i < r->devices.size() && rc == RPF_OK; ++i) {
        hipError_t err = hipSetDevice(r->devices[i]);
        if (err == hipSuccess && i == 0)
            err = hipMemcpyAsync(host_out, r->d_result, sizeof(double) * count, hipMemcpyDeviceToHost, r->streams[0]);
        if (err == hipSuccess) err = hipStreamSynchronize(r->streams[i]);
        if (err != hipSuccess) rc = rfail(r, RPF_ERR_HARDWARE, std::string("rpf_scan_reducer_reduce: ") + hipGetErrorString(err));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
