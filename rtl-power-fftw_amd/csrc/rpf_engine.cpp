// rpf_engine.cpp -- implementation of include/rpf_engine.h: the buffer pool and
// consumer thread of Datastore (/root/reference/src/datastore.cxx:23-103)
// re-designed around a HIP device.
//
//   producer (caller's thread)                consumer (engine thread)                  recycler (engine thread)
//   --------------------------                ------------------------                  ------------------------
//   rpf_buffer_acquire  <-- empty_  <-------------------------------------------------  each buffer as soon as ITS
//   fill pinned buffer                                                                   H2D copy has landed
//   rpf_buffer_submit   --> occupied_ ------->  pop everything queued; per buffer one
//                                               hipMemcpyAsync H2D, at once, on
//                                               alternating copy streams, into the
//                                               device staging slot being filled; when
//                                               the slot is full (or at the end): carry
//                                               the previous slot's unfinished frame in
//                                               front, fused kernel + reduce (compute stream)
//   rpf_finish          --> finished_ ------->  launch the last slot, sync, pwr -> host, exit
//
// Pinned host buffers let the H2D copies overlap the kernels (copy streams + compute stream + events); a frame that
// straddles two slots (datastore.cxx:52,68,81) is completed by copying the tail of the previous staging slot in front
// of the new bytes, device to device.
#include "../../include/rpf_engine.h"
#include "rpf_engine_testing.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bluestein_tables.h"
#include "rpf_kernels.h"

namespace {

thread_local std::string g_last_error;

struct HostBuffer {
    uint8_t* data = nullptr;
    size_t size = 0;   // bytes valid after submit (Buffer::size())
    hipEvent_t copied = nullptr;    // its H2D copy has finished: the producer may refill it
    bool external = false;          // a piece of a caller's registered stream (rpf_accumulate's direct path): never recycled
};

constexpr int kStagingSlots = 3;    // device staging ring: one slot filling, one waiting for its kernel, one running
constexpr int kCopyStreams = 2;     // H2D copies alternate between two streams (two SDMA queues): no gap between copies

struct StagingSlot {
    uint8_t* base = nullptr;        // device memory: [head_room | coalesce x buffer_capacity]
    hipEvent_t copy_done[kCopyStreams] = {};   // everything copied into the slot on that stream has landed
    hipEvent_t kernel_done = nullptr;
    bool in_flight = false;
    // the last transform launched over the slot (still valid until the slot is reopened): what a fused launch that
    // gave up is re-run on
    const uint8_t* launch_ptr = nullptr;
    int64_t launch_frames = 0;
    bool launched_fused = false;
    unsigned* verdict = nullptr;    // pinned host word the launch's verdict kernel sets to 1 if it gave up
};

}  // namespace

struct rpf_engine {
    // configuration (Params fields Datastore reads)
    int N = 0;
    bool has_window = false;
    int n_buffers = 0;
    size_t buffer_capacity = 0;
    int device = 0;
    uint32_t flags = 0;
    bool use_dma = true;
    int variant = 0;

    // Datastore public state
    int64_t repeats = 0;        // params.repeats of the running acquisition
    int64_t repeats_done = 0;   // datastore.h:38
    std::mutex status_mutex;    // datastore.h:40
    std::deque<HostBuffer*> empty_buffers;      // :43
    std::deque<HostBuffer*> occupied_buffers;   // :44
    bool acquisition_finished = false;          // :45
    std::condition_variable status_change;      // :46
    std::vector<int> queue_histogram;           // :47
    std::vector<double> pwr;                    // :53

    std::vector<HostBuffer> pool;
    uint8_t* pool_base = nullptr;        // the pool as ONE pinned allocation: neighbouring buffers can travel in one copy
    std::vector<std::pair<const uint8_t*, size_t>> registered;     // rpf_stream_register: caller memory pinned for direct replay
    size_t bytes_landed = 0;             // H2D bytes whose copy has finished (recycler; under recycle_mutex)
    std::thread worker;
    bool worker_running = false;
    int worker_rc = RPF_OK;
    std::string worker_error;

    // device side
    hipStream_t copy_streams[kCopyStreams] = {}, compute_stream = nullptr;
    // buffers whose copies are in flight, in issue order; the recycler thread waits for each copy and hands the buffer back
    std::mutex recycle_mutex;
    std::condition_variable recycle_cv;
    std::deque<std::pair<hipEvent_t, HostBuffer*>> recycle_queue;
    bool recycle_stop = false;
    std::string recycler_error;
    rpf::cf* d_twiddles = nullptr;
    bool fourstep = false;                // N handled by rpf_fourstep.hip
    bool fused = false;                   // ... by the fused persistent kernel (Y stays in the XCDs' L2)
    void* d_fused_ctl = nullptr;          // its team counters / abort flag
    rpf::cf* d_fused_scratch = nullptr;   // its Y (two rounds per XCD); kept apart from d_scratch, the two-kernel path's
    // What became of the fused launches, written by their verdict kernels into pinned host memory:
    // [0 .. 2] one word per staging slot (queue path), [3] launches of the device-resident entries that gave up
    // (cumulative), [4] all launches that gave up (cumulative).
    unsigned* h_fused_words = nullptr;
    unsigned* d_fused_words = nullptr;    // the same words as the device addresses them
    bool last_was_fused = false;          // the last transform launched was the fused kernel
    unsigned fused_aborts_seen = 0;       // h_fused_words[3] when the host last looked
    int64_t fused_recovered = 0;          // launches re-run on K2a/K2b after giving up (queue path)
    int fused_fault_mode = 0, fused_fault_skip = 0, fused_fault_count = 0;   // rpf_debug_fused_fault
    bool bluestein = false;               // N handled by the Bluestein kernel (chirp tables below)
    bool bigblu = false;                  // N handled by the large (four-step) Bluestein path
    bool mixed = false;                   // N handled by the LDS mixed-radix kernel (rpf_mixed.hip)
    bool generic = false;                 // N handled by the catch-all Stockham path (rpf_generic.hip)
    int gen_h = 0;                        // its two-level twiddle split (d_tw_sub = T0, d_tw_sub2 = T1)
    int blu_M = 0;                        // bigblu: convolution length (partial spectra have M entries)
    rpf::cf* d_chirp = nullptr;           // g[n], N entries
    rpf::cf* d_bhat = nullptr;            // frequency-domain chirp, M entries
    rpf::cf* d_tw_sub = nullptr;          // four-step: twiddles of the N1-point column transforms
    rpf::cf* d_tw_sub2 = nullptr;         // four-step: twiddles of the N2-point row transforms
    rpf::cf* d_scratch = nullptr;         // four-step: intermediate Y
    size_t scratch_bytes = 0;             // its size now; grown on demand (two-kernel four-step and large Bluestein paths)
    size_t scratch_per_frame = 0, scratch_max = 0;
    rpf::cf* d_step2 = nullptr;           // large Bluestein: second transform's inter-step twiddles
    float* d_window = nullptr;
    double* d_partial = nullptr;
    double* d_pwr = nullptr;
    size_t head_room = 0;                 // >= 2N, multiple of 256
    size_t coalesce = 1;                  // host buffers one staging slot holds (= one transform launch)
    std::vector<StagingSlot> staging;
    rpf::LaunchInfo plan;                 // resident grid for this N
    rpf::LaunchInfo last;                 // last launch
    int last_slots = 0;                   // partial spectra left by the last transform
    rpf::SlotRanges last_ranges;          // ... and which of them belong to which hop (K1 hop launches)
    int last_hops = 0;                    // hops of the last rpf_device_fused_hops (0: a single-acquisition transform)

    mutable std::string last_error;
};

namespace {

// Waits on `cv` until `pred` holds, after first polling for it for up to ~100 us with the lock dropped between looks.
// The three hand-offs of a buffer's round trip (copy landed -> recycler -> producer -> consumer) are each a thread
// wake-up; a condition-variable wake-up costs 10 - 50 us and a 1.6 MB buffer is 30 us of PCIe time, so with the
// reference's five buffers sleeping threads left the link idle a quarter of the time (41 of the 53.7 GB/s two copy
// streams reach, tools/h2d_rate.cpp).  A waiter that has just been busy looks again before it sleeps.
template <class Pred>
void wait_briefly_then_block(std::unique_lock<std::mutex>& lock, std::condition_variable& cv, Pred pred)
{
    if (pred()) return;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(100);
    do {
        lock.unlock();
        std::this_thread::yield();
        lock.lock();
        if (pred()) return;
    } while (std::chrono::steady_clock::now() < deadline);
    cv.wait(lock, pred);
}

int fail(rpf_engine* e, int rc, const std::string& msg)
{
    if (e) e->last_error = msg;
    g_last_error = msg;
    return rc;
}

// Makes the engine's device current for the scope of an entry point and puts the
// caller's back afterwards: in a multi-GPU process (one engine per device, or a
// caller that has another device selected) a launch against engine-owned memory
// must not depend on -- nor disturb -- the calling thread's current device.
class DeviceScope {
public:
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        status_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
        changed_ = (status_ == hipSuccess && prev_ != device);
    }
    ~DeviceScope()
    {
        if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    hipError_t status() const { return status_; }
private:
    int prev_ = -1;
    bool changed_ = false;
    hipError_t status_ = hipSuccess;
};

#define HIP_TRY(e, call)                                                                 \
    do {                                                                                 \
        hipError_t err__ = (call);                                                       \
        if (err__ != hipSuccess)                                                         \
            return fail(e, RPF_ERR_HARDWARE,                                             \
                        std::string(#call) + ": " + hipGetErrorString(err__));           \
    } while (0)

// K1 over up to rpf::kMaxHops acquisitions in ONE launch (hop_partition.h): leaves the partial
// spectra of hop h in the slots [slots->begin[h], slots->begin[h+1]) of e->d_partial.
int launch_fused_hops(rpf_engine* e, const uint8_t* const* d_frames, const int64_t* nframes, int H,
                      hipStream_t stream, rpf::SlotRanges* slots, int* nslots)
{
    if (H == 1
#ifdef RPF_TUNING
        && !std::getenv("RPF_TUNE_SCAN_KERNEL")
#endif
    ) {
        // a scan of one hop is a single acquisition: the plain kernel (1.8 us per launch less)
        for (int h = 0; h <= rpf::kMaxHops; ++h) slots->begin[h] = 0;
        *nslots = 0;
        if (nframes[0] < 1) return RPF_OK;
        const int64_t wanted = (nframes[0] + e->plan.fpw - 1) / e->plan.fpw;
        const int grid = static_cast<int>(std::min<int64_t>(e->plan.grid, wanted));
        const bool dma1 = e->use_dma && (reinterpret_cast<uintptr_t>(d_frames[0]) % 16) == 0;
        HIP_TRY(e, rpf::launch_fft_accum(e->N, e->variant, e->has_window, dma1, d_frames[0], nframes[0], e->d_twiddles,
                                         e->d_window, e->d_partial, grid, stream, &e->last));
        for (int h = 1; h <= rpf::kMaxHops; ++h) slots->begin[h] = grid;
        *nslots = grid;
        return RPF_OK;
    }
    rpf::HopArgs args;
    bool interleave_single = false;
#ifdef RPF_TUNING
    interleave_single = H == 1 && std::getenv("RPF_TUNE_INTERLEAVE") != nullptr;     // A/B of the two iteration orders
#endif
    const int grid = rpf::partition_hops(nframes, H, e->plan.fpw, e->plan.grid, &args, slots, interleave_single);
    if (grid < 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "too many frames for one launch");
    bool dma = e->use_dma;
    for (int h = 0; h < rpf::kMaxHops; ++h) {
        args.stream[h] = h < H ? d_frames[h] : nullptr;
        if (h < H && nframes[h] > 0 && (reinterpret_cast<uintptr_t>(d_frames[h]) % 16) != 0) dma = false;
    }
    *nslots = slots->begin[H];
    if (grid == 0) return RPF_OK;           // no whole frame anywhere: every slot range is empty
    HIP_TRY(e, rpf::launch_fft_accum_hops(e->N, e->variant, e->has_window, dma, args, e->d_twiddles, e->d_window,
                                          e->d_partial, grid, stream, &e->last));
    return RPF_OK;
}

// The intermediate of the two-kernel four-step and large Bluestein paths: sized to what the launches need, not to the
// 2 GB (4 GB) a whole device-resident acquisition can use -- a queue-fed engine never launches more than a 32 MB slot
// of input.  Grows (never shrinks) when a larger launch arrives: one stream synchronisation, then free + allocate; if
// the device cannot give more the launch runs in smaller batches on what there is.
int ensure_scratch(rpf_engine* e, int64_t nframes, hipStream_t stream)
{
    if (!e->scratch_per_frame) return RPF_OK;
    const size_t want = std::min<size_t>(e->scratch_max, static_cast<size_t>(nframes) * e->scratch_per_frame);
    if (want <= e->scratch_bytes) return RPF_OK;
    HIP_TRY(e, hipStreamSynchronize(stream));
    HIP_TRY(e, hipStreamSynchronize(e->compute_stream));
    const size_t had = e->scratch_bytes;
    (void)hipFree(e->d_scratch);
    e->d_scratch = nullptr;
    e->scratch_bytes = 0;
    void* p = nullptr;
    if (hipMalloc(&p, want) == hipSuccess) {
        e->scratch_bytes = want;
    } else {
        (void)hipGetLastError();
        // what there was before, or (nothing yet: an engine that has just left the fused kernel) one frame's worth
        const size_t least = std::max(had, e->scratch_per_frame);
        HIP_TRY(e, hipMalloc(&p, least));
        e->scratch_bytes = least;
    }
    e->d_scratch = static_cast<rpf::cf*>(p);
    return RPF_OK;
}

// This engine keeps to K2a/K2b from here on (a fused launch gave up).  Nothing is freed -- launches in flight may
// still be using the fused kernel's scratch --; the two-kernel path's intermediate is allocated by ensure_scratch at
// its first launch (one stream synchronisation).
void retire_fused(rpf_engine* e)
{
    if (!e->fused) return;
    e->fused = false;
    e->scratch_per_frame = rpf::fourstep_scratch_bytes_per_frame(e->N);
    e->scratch_max = rpf::fourstep_scratch_bytes(e->N);
}

// The device-resident entries return without synchronising, so a launch of theirs that gave up is found out later:
// its verdict kernel has counted it in pinned host memory by the time the caller's stream has passed it.  Every
// later entry looks (no synchronisation) and retires the fused kernel once it sees a count it has not seen.
void note_device_path_aborts(rpf_engine* e)
{
    if (!e->h_fused_words) return;
    const unsigned now = __atomic_load_n(&e->h_fused_words[3], __ATOMIC_ACQUIRE);
    if (now != e->fused_aborts_seen) {
        e->fused_aborts_seen = now;
        retire_fused(e);
    }
}

// Enqueue K1 (or the four-step pair K2a/K2b) for `nframes` frames starting at
// d_frames; leaves *nslots partial spectra in e->d_partial.
int launch_transform(rpf_engine* e, const uint8_t* d_frames, int64_t nframes, hipStream_t stream,
                     int* nslots)
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_frames);
    e->last_was_fused = false;
    if (e->fourstep && e->fused) {
        const bool dma = e->use_dma && (addr % 4) == 0;
        int fault = 0;                       // rpf_debug_fused_fault: `count` launches after `skip` untouched ones
        if (e->fused_fault_count != 0) {
            if (e->fused_fault_skip > 0) {
                --e->fused_fault_skip;
            } else {
                fault = e->fused_fault_mode;
                if (e->fused_fault_count > 0) --e->fused_fault_count;
            }
        }
        HIP_TRY(e, rpf::launch_fourstep_fused(e->N, e->has_window, dma, d_frames, nframes, e->d_tw_sub, e->d_tw_sub2,
                                              e->d_twiddles, e->d_window, e->d_fused_scratch, e->d_partial,
                                              e->d_fused_ctl, stream, fault));
        e->last = e->plan;
        e->last_was_fused = true;
        *nslots = rpf::fourstep_fused_slots(e->N);
        return RPF_OK;
    }
    if (e->fourstep) {
        const bool dma = e->use_dma && (addr % 4) == 0;
        if (int rc = ensure_scratch(e, nframes, stream)) return rc;
        HIP_TRY(e, rpf::launch_fourstep(e->N, e->has_window, dma, d_frames, nframes, e->d_tw_sub,
                                        e->d_tw_sub2, e->d_twiddles, e->d_window, e->d_scratch, e->scratch_bytes,
                                        e->d_partial, e->plan.grid, stream));
        e->last = e->plan;
        *nslots = rpf::fourstep_partial_slots(e->N);
        return RPF_OK;
    }
    if (e->generic) {
        HIP_TRY(e, rpf::launch_generic(e->N, d_frames, nframes, e->d_window, e->d_chirp, e->d_bhat, e->d_tw_sub,
                                       e->d_tw_sub2, e->gen_h, e->d_scratch, e->d_partial, /*accumulate=*/false, stream));
        e->last = e->plan;
        *nslots = 1;
        return RPF_OK;
    }
    if (e->bigblu) {
        const bool dma = e->use_dma && (addr % 4) == 0;
        if (int rc = ensure_scratch(e, nframes, stream)) return rc;
        HIP_TRY(e, rpf::launch_bigblu(e->N, dma, d_frames, nframes, e->d_tw_sub, e->d_tw_sub2, e->d_twiddles,
                                      e->d_step2, e->d_chirp, e->d_bhat, e->d_scratch, e->scratch_bytes, e->d_partial,
                                      e->plan.grid, stream));
        e->last = e->plan;
        *nslots = rpf::bigblu_partial_slots(e->N);
        return RPF_OK;
    }
    if (e->mixed) {       // (trims the grid to the frames itself: the split form needs a multiple of its factor)
        HIP_TRY(e, rpf::launch_mixed(e->N, e->variant, d_frames, nframes, e->d_twiddles, e->d_window, e->d_partial,
                                     e->plan.grid, stream, &e->last));
        *nslots = e->last.slots ? e->last.slots : e->last.grid;
        return RPF_OK;
    }
    if (e->bluestein) {
        const int64_t wanted = (nframes + e->plan.fpw - 1) / e->plan.fpw;
        const int grid = static_cast<int>(std::min<int64_t>(e->plan.grid, wanted));
        HIP_TRY(e, rpf::launch_bluestein(e->N, d_frames, nframes, e->d_twiddles, e->d_chirp, e->d_bhat,
                                         e->d_partial, grid, stream, &e->last));
        *nslots = grid;
        return RPF_OK;
    }
    // K1, one acquisition per launch
    // (RPF_TUNE_SCAN_KERNEL, tuning build: the scan kernel on a scan of one hop, A/B)
    rpf::SlotRanges slots;
    (void)addr;
    return launch_fused_hops(e, &d_frames, &nframes, 1, stream, &slots, nslots);
}

// Transform + reduce for `nframes` frames starting at d_frames.
// slot_verdict: the queue path's pinned word for this launch -- if a fused launch gives up, K3 leaves d_out as it
// is, the word becomes 1 and the worker re-runs the bytes on K2a/K2b (recover_fused); null (the device-resident
// entries): d_out is NaN-filled and the launch counted in h_fused_words[3].
int launch_frames(rpf_engine* e, const uint8_t* d_frames, int64_t nframes, double* d_out,
                  bool accumulate, hipStream_t stream, unsigned* slot_verdict = nullptr)
{
    if (nframes <= 0) return RPF_OK;
    if (e->fused && !slot_verdict) note_device_path_aborts(e);
    int nslots = 0;
    int rc = launch_transform(e, d_frames, nframes, stream, &nslots);
    if (rc != RPF_OK) return rc;
    e->last_slots = nslots;
    e->last_hops = 0;
    const bool fused = e->last_was_fused;
    HIP_TRY(e, rpf::launch_reduce(e->d_partial, nslots, e->N, d_out, accumulate, stream,
                                  e->plan.partial_f32, e->bigblu ? static_cast<size_t>(e->blu_M) : 0,
                                  fused && slot_verdict ? rpf::fourstep_fused_abort_word(e->d_fused_ctl) : nullptr));
    // a fused launch whose teams did not assemble must not leave something that looks like a spectrum
    if (fused)
        HIP_TRY(e, rpf::launch_fused_verdict(e->d_fused_ctl, slot_verdict ? nullptr : d_out, e->N,
                                             slot_verdict ? slot_verdict : e->d_fused_words + 3, e->d_fused_words + 4, stream));
    return RPF_OK;
}

// Hands the pool's buffers back to the producer, each as soon as ITS copy has landed (datastore.cxx:91-94 returns a
// buffer when the worker is done with it; here "done" is the end of the H2D copy, not of the transform).  A thread of
// its own so that the consumer thread never blocks on a copy while new buffers are waiting to be issued: with five
// 1.6 MB buffers a blocking consumer alternated 4-buffer and 1-buffer groups and left the link idle between them
// (37 GB/s of the 54.5 GB/s large buffers reach).
void recycler_main(rpf_engine* e)
{
    (void)hipSetDevice(e->device);
    std::unique_lock<std::mutex> lk(e->recycle_mutex);
    for (;;) {
        wait_briefly_then_block(lk, e->recycle_cv, [&]() { return !e->recycle_queue.empty() || e->recycle_stop; });
        if (e->recycle_queue.empty()) break;                     // stop requested and everything handed back
        const std::pair<hipEvent_t, HostBuffer*> item = e->recycle_queue.front();
        e->recycle_queue.pop_front();
        lk.unlock();
        if (item.first) {
            const hipError_t err = hipEventSynchronize(item.first);
            if (err != hipSuccess && e->recycler_error.empty())
                e->recycler_error = std::string("hipEventSynchronize(copied): ") + hipGetErrorString(err);
        }
        const size_t landed = item.first ? item.second->size : 0;
        {
            std::lock_guard<std::mutex> status(e->status_mutex);
            e->empty_buffers.push_back(item.second);             // datastore.cxx:91-94
            e->status_change.notify_all();
        }
        lk.lock();
        e->bytes_landed += landed;
    }
}

// Consumer thread: the GPU counterpart of Datastore::fftThread.
void worker_main(rpf_engine* e)
{
    // First failure wins; after it the worker issues no further HIP work and only
    // keeps the hand-off protocol alive (buffers go straight back to the producer)
    // until rpf_finish reports the error.
    auto bail = [&](hipError_t err, const char* what) {
        if (e->worker_rc != RPF_OK) return;
        e->worker_rc = RPF_ERR_HARDWARE;
        e->worker_error = std::string(what) + ": " + hipGetErrorString(err);
    };
    auto ok = [&]() { return e->worker_rc == RPF_OK; };
#define WORKER_TRY(call, what)                          \
    do {                                                \
        if (ok()) {                                     \
            hipError_t err__ = (call);                  \
            if (err__ != hipSuccess) bail(err__, what); \
        }                                               \
    } while (0)
    WORKER_TRY(hipSetDevice(e->device), "hipSetDevice");
    {
        std::lock_guard<std::mutex> lk(e->recycle_mutex);
        e->recycle_stop = false;
        e->recycler_error.clear();
        e->bytes_landed = 0;
    }
    std::thread recycler;
    try {
        recycler = std::thread(recycler_main, e);
    } catch (const std::exception& ex) {
        // no second thread to be had: the acquisition fails (rpf_finish says why) and this thread keeps the hand-off
        // protocol alive by returning every buffer itself
        if (ok()) {
            e->worker_rc = RPF_ERR_HARDWARE;
            e->worker_error = std::string("cannot start the buffer recycler thread: ") + ex.what();
        }
    }
    auto hand_back = [&](HostBuffer* b, hipEvent_t copied) {
        if (!recycler.joinable()) {
            if (copied) (void)hipEventSynchronize(copied);
            std::lock_guard<std::mutex> status(e->status_mutex);
            e->empty_buffers.push_back(b);
            e->status_change.notify_all();
            return;
        }
        std::lock_guard<std::mutex> lk(e->recycle_mutex);
        e->recycle_queue.emplace_back(copied, b);
        e->recycle_cv.notify_one();
    };

    const size_t frame_bytes = 2 * static_cast<size_t>(e->N);
    const size_t slot_bytes = e->coalesce * e->buffer_capacity;
    size_t carry = 0;             // bytes of an unfinished frame at the end of the previous slot
    const uint8_t* carry_src = nullptr;
    size_t slot_idx = 0;
    int64_t frames_issued = 0;    // == repeats_done once everything has drained
    // The staging slot being filled: buffers are copied into it one by one as the producer submits them -- each copy
    // issued at once, on alternating copy streams -- and ONE transform is launched over the slot when it is full or the
    // acquisition ends; the stream the kernels see is the concatenation of the buffers in FIFO order either way.
    StagingSlot* cur = nullptr;
    size_t off = 0;               // bytes copied into `cur`
    bool used[kCopyStreams] = {};
    unsigned next_stream = 0;
    size_t bytes_issued = 0;      // H2D bytes whose copy has been issued (against rpf_engine::bytes_landed)

    // A fused four-step launch gave up (its teams did not assemble: rpf_fourstep.hip): K3 has left d_pwr alone and the
    // slot still holds the launch's bytes.  Everything in flight is waited for, the engine leaves the fused kernel for
    // good, and every slot whose launch gave up is run again on K2a/K2b, oldest first -- the order of the additions
    // into d_pwr is the launch order whenever all fused launches in flight gave up (a device that has become busy),
    // and differs from it by the place of the re-run terms otherwise (double addition: ~1e-16 relative).
    // The reference's worker has no failure path (datastore.cxx:48-96); neither has this one for this cause.
    auto recover_fused = [&](size_t oldest) {
        WORKER_TRY(hipStreamSynchronize(e->compute_stream), "hipStreamSynchronize(recover)");
        retire_fused(e);
        for (size_t k = 0; k < e->staging.size(); ++k) {
            StagingSlot& s = e->staging[(oldest + k) % e->staging.size()];
            if (!s.launched_fused || !s.verdict || __atomic_load_n(s.verdict, __ATOMIC_ACQUIRE) == 0) continue;
            *s.verdict = 0;
            s.launched_fused = false;
            if (!ok()) continue;
            int rc = launch_frames(e, s.launch_ptr, s.launch_frames, e->d_pwr, /*accumulate=*/true, e->compute_stream);
            if (rc != RPF_OK) {
                e->worker_rc = rc;
                e->worker_error = e->last_error;
            }
            ++e->fused_recovered;
        }
        // (the re-runs read the slots: nothing may be copied into one before they are done)
        WORKER_TRY(hipStreamSynchronize(e->compute_stream), "hipStreamSynchronize(recover)");
        for (auto& s : e->staging) s.in_flight = false;
    };
    auto gave_up = [&](const StagingSlot& s) {
        return s.launched_fused && s.verdict && __atomic_load_n(s.verdict, __ATOMIC_ACQUIRE) != 0;
    };
    auto open_slot = [&]() {
        const size_t idx = slot_idx;
        cur = &e->staging[slot_idx];
        slot_idx = (slot_idx + 1) % e->staging.size();
        // The slot is free once its own kernel AND the following slot's carry copy (which read its tail) are done;
        // the latter precedes that slot's kernel_done in stream order.
        StagingSlot& after = e->staging[slot_idx];
        for (StagingSlot* s : {cur, &after}) {
            if (!s->in_flight) continue;
            WORKER_TRY(hipEventSynchronize(s->kernel_done), "hipEventSynchronize(kernel_done)");
            s->in_flight = false;
            if (gave_up(*s)) recover_fused(idx);        // (`cur` is the oldest slot in flight)
        }
        cur->launched_fused = false;
        off = 0;
        for (bool& u : used) u = false;
    };
    auto launch_slot = [&]() {
        if (!cur) return;
        // new bytes sit right after the head room; the carried partial frame goes immediately in front of them
        uint8_t* const dst = cur->base + e->head_room;
        for (int c = 0; c < kCopyStreams; ++c) {
            if (!used[c]) continue;
            WORKER_TRY(hipEventRecord(cur->copy_done[c], e->copy_streams[c]), "hipEventRecord(copy_done)");
            WORKER_TRY(hipStreamWaitEvent(e->compute_stream, cur->copy_done[c], 0), "hipStreamWaitEvent");
        }
        if (carry)
            WORKER_TRY(hipMemcpyAsync(dst - carry, carry_src, carry, hipMemcpyDeviceToDevice, e->compute_stream),
                       "hipMemcpyAsync(carry)");
        const size_t avail = carry + off;
        int64_t nframes = static_cast<int64_t>(avail / frame_bytes);
        nframes = std::min<int64_t>(nframes, e->repeats - frames_issued);    // datastore.cxx:67
        if (ok() && nframes > 0) {
            if (cur->verdict) *cur->verdict = 0;
            int rc = launch_frames(e, dst - carry, nframes, e->d_pwr, /*accumulate=*/true, e->compute_stream, cur->verdict);
            if (rc != RPF_OK) {
                e->worker_rc = rc;
                e->worker_error = e->last_error;
            }
            cur->launch_ptr = dst - carry;
            cur->launch_frames = nframes;
            cur->launched_fused = rc == RPF_OK && e->last_was_fused;
            frames_issued += nframes;
        }
        WORKER_TRY(hipEventRecord(cur->kernel_done, e->compute_stream), "hipEventRecord(kernel_done)");
        cur->in_flight = ok();
        // the unfinished frame (if any) stays in this slot until the next one is launched
        const size_t consumed = static_cast<size_t>(std::max<int64_t>(nframes, 0)) * frame_bytes;
        carry = (frames_issued < e->repeats) ? (avail - consumed) % frame_bytes : 0;
        carry_src = dst + off - carry;
        cur = nullptr;
        off = 0;
    };

    std::vector<HostBuffer*> batch;
    std::unique_lock<std::mutex> status_lock(e->status_mutex, std::defer_lock);
    while (true) {
        // Wait until we have a bufferful of data (datastore.cxx:54-64)
        status_lock.lock();
        wait_briefly_then_block(status_lock, e->status_change,
                                [&]() { return !e->occupied_buffers.empty() || e->acquisition_finished; });
        if (e->occupied_buffers.empty()) {
            status_lock.unlock();
            break;   // acquisition finished
        }
        batch.assign(e->occupied_buffers.begin(), e->occupied_buffers.end());
        e->occupied_buffers.clear();
        status_lock.unlock();

        // A lone buffer while the link is busy anyway: look a few microseconds longer for its neighbour -- two buffers that
        // lie next to each other in the pool travel in ONE copy (below), and a 1.6 MB copy pays ~4 us of set-up per 29 us of
        // transfer (tools/h2d_rate.cpp: 50 GB/s in 1.6 MB copies, 57 GB/s in 6.5 MB ones).  Costs nothing: at least two
        // buffers' worth of bytes are still on their way.
        if (batch.size() == 1 && ok() && !batch[0]->external) {
            size_t landed;
            {
                std::lock_guard<std::mutex> lk(e->recycle_mutex);
                landed = e->bytes_landed;
            }
            const size_t in_flight = bytes_issued - landed;
            if (in_flight >= 2 * e->buffer_capacity) {
                // (the more is on its way, the longer the look may take: a buffer's worth is ~30 us of link time)
                const auto deadline = std::chrono::steady_clock::now() +
                                      std::chrono::microseconds(in_flight >= 3 * e->buffer_capacity ? 20 : 8);
                do {
                    std::this_thread::yield();
                    status_lock.lock();
                    if (!e->occupied_buffers.empty()) {
                        batch.insert(batch.end(), e->occupied_buffers.begin(), e->occupied_buffers.end());
                        e->occupied_buffers.clear();
                    }
                    status_lock.unlock();
                } while (batch.size() == 1 && std::chrono::steady_clock::now() < deadline);
            }
        }

        for (size_t i = 0; i < batch.size();) {
            HostBuffer* const b = batch[i];
            // datastore.cxx:67: once the quota is met (by what is staged already) the rest of the stream is ignored
            const int64_t staged = frames_issued + static_cast<int64_t>((carry + off) / frame_bytes);
            if (!ok() || b->size == 0 || staged >= e->repeats) {
                if (!b->external) hand_back(b, nullptr);
                ++i;
                continue;
            }
            if (cur && off + b->size > slot_bytes) launch_slot();
            if (!cur) open_slot();
            // the run of buffers that follow b in the queue AND in memory (the pool is one allocation; the pieces of a
            // registered stream are consecutive anyway), as far as the slot and the quota take them: one copy
            size_t run = 1, bytes = b->size;
            while (i + run < batch.size()) {
                const HostBuffer* const nb = batch[i + run];
                if (nb->data != b->data + bytes || nb->size == 0 || nb->external != b->external || off + bytes + nb->size > slot_bytes) break;
                if (frames_issued + static_cast<int64_t>((carry + off + bytes) / frame_bytes) >= e->repeats) break;
                bytes += nb->size;
                ++run;
            }
            const unsigned c = next_stream;
            next_stream = (next_stream + 1) % kCopyStreams;
            WORKER_TRY(hipMemcpyAsync(cur->base + e->head_room + off, b->data, bytes, hipMemcpyHostToDevice, e->copy_streams[c]),
                       "hipMemcpyAsync(H2D)");
            HostBuffer* const last = batch[i + run - 1];
            if (!b->external) WORKER_TRY(hipEventRecord(last->copied, e->copy_streams[c]), "hipEventRecord(copied)");
            used[c] = true;
            off += bytes;
            bytes_issued += bytes;
            if (!b->external)
                for (size_t k = 0; k < run; ++k) hand_back(batch[i + k], ok() ? last->copied : nullptr);
            i += run;
        }
    }
    launch_slot();   // what the last, partly filled slot holds

    {
        std::lock_guard<std::mutex> lk(e->recycle_mutex);
        e->recycle_stop = true;
        e->recycle_cv.notify_one();
    }
    if (recycler.joinable()) recycler.join();             // every buffer is back in empty_buffers
    if (ok() && !e->recycler_error.empty()) {
        e->worker_rc = RPF_ERR_HARDWARE;
        e->worker_error = e->recycler_error;
    }

    WORKER_TRY(hipStreamSynchronize(e->compute_stream), "hipStreamSynchronize");
    // the launches still in flight at the end: any fused one that gave up is re-run now (slot_idx = the oldest slot)
    for (const auto& s : e->staging)
        if (ok() && gave_up(s)) {
            recover_fused(slot_idx);
            break;
        }
    for (auto& s : e->staging) {
        s.in_flight = false;
        s.launched_fused = false;
    }
    WORKER_TRY(hipMemcpy(e->pwr.data(), e->d_pwr, sizeof(double) * e->N, hipMemcpyDeviceToHost), "hipMemcpy(pwr)");
    e->repeats_done = frames_issued;
#undef WORKER_TRY
}

void release_device(rpf_engine* e)
{
    if (e->d_twiddles) (void)hipFree(e->d_twiddles);
    if (e->d_tw_sub) (void)hipFree(e->d_tw_sub);
    if (e->d_tw_sub2) (void)hipFree(e->d_tw_sub2);
    if (e->d_scratch) (void)hipFree(e->d_scratch);
    if (e->d_fused_ctl) (void)hipFree(e->d_fused_ctl);
    if (e->d_fused_scratch) (void)hipFree(e->d_fused_scratch);
    if (e->h_fused_words) (void)hipHostFree(e->h_fused_words);
    if (e->d_step2) (void)hipFree(e->d_step2);
    if (e->d_chirp) (void)hipFree(e->d_chirp);
    if (e->d_bhat) (void)hipFree(e->d_bhat);
    if (e->d_window) (void)hipFree(e->d_window);
    if (e->d_partial) (void)hipFree(e->d_partial);
    if (e->d_pwr) (void)hipFree(e->d_pwr);
    for (auto& s : e->staging) {
        if (s.base) (void)hipFree(s.base);
        if (s.kernel_done) (void)hipEventDestroy(s.kernel_done);
        for (hipEvent_t ev : s.copy_done)
            if (ev) (void)hipEventDestroy(ev);
    }
    for (hipStream_t cs : e->copy_streams)
        if (cs) (void)hipStreamDestroy(cs);
    if (e->compute_stream) (void)hipStreamDestroy(e->compute_stream);
    for (auto& b : e->pool) {
        if (b.data && !e->pool_base) (void)hipHostFree(b.data);
        if (b.copied) (void)hipEventDestroy(b.copied);
    }
    if (e->pool_base) (void)hipHostFree(e->pool_base);
    for (const auto& r : e->registered) (void)hipHostUnregister(const_cast<uint8_t*>(r.first));
}

}  // namespace

extern "C" {

int rpf_abi_version(void) { return RPF_ABI_VERSION; }

int rpf_supported_n(int N)
{
    return (rpf::kernel_supported(N) || rpf::fourstep_supported(N) || rpf::mixed_supported(N) || rpf::bluestein_supported(N) ||
            rpf::bigblu_supported(N) || rpf::generic_supported(N)) ? 1 : 0;
}

const char* rpf_last_global_error(void) { return g_last_error.c_str(); }

int rpf_engine_create(const rpf_config* cfg, rpf_engine** out)
{
    if (!out) return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_engine_create: out is NULL");
    *out = nullptr;
    if (!cfg || cfg->struct_size != sizeof(rpf_config))
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_engine_create: bad rpf_config size");
    if (cfg->N < 2 || (cfg->N % 2) != 0)
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT,
                    "Number of bins must be a positive even number.");
    const int variant = static_cast<int>((cfg->flags >> 8) & 0xffu);
    // (asking for the fused four-step kernel is asking for the four-step path)
    // 32768 is served twice, by the split form 2 x 16384 and by the four-step kernels.  Plain runs are faster on the
    // former (0.37 against 0.22 Tsample/s); WINDOWED runs are not: the 16384-point plan has no registers left for the
    // window values next to its two-deep section pipeline (0.175, it was 0.27 before round 3's section sums), the
    // four-step kernels multiply them in as they unpack (0.21 - 0.22).  profiles/r04_sizes.txt.
    const bool windowed_32768 = cfg->N == 32768 && cfg->window != nullptr && variant == 0;
    const bool mixed = rpf::mixed_supported(cfg->N, variant) && !windowed_32768 &&
                       !(cfg->flags & (RPF_FLAG_NO_MIXED_RADIX | RPF_FLAG_FOURSTEP_FUSED));
    const bool fourstep = !mixed && rpf::fourstep_supported(cfg->N) && variant == 0;
    const bool bluestein = !mixed && rpf::bluestein_supported(cfg->N) && variant == 0;
    const bool bigblu = !mixed && rpf::bigblu_supported(cfg->N) && variant == 0;
    const bool tuned = fourstep || mixed || bluestein || bigblu || rpf::kernel_supported(cfg->N, variant);
    const bool generic = !tuned && variant == 0 && rpf::generic_supported(cfg->N);
    if (!tuned && !generic)
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT,
                    "No gfx950 kernel for " + std::to_string(cfg->N) +
                        " bins in this build (supported: every even N up to 8388608 and the powers of two up to 67108864).");
    if (cfg->n_buffers < 1)
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "Argument to 'buffers' must be a positive number.");
    if (cfg->buffer_capacity < 2 || (cfg->buffer_capacity % 2) != 0)
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "Buffer size must be a positive even number of bytes.");

    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev < 1)
        return fail(nullptr, RPF_ERR_HARDWARE,
                    std::string("No HIP device available (") +
                        (err != hipSuccess ? hipGetErrorString(err) : "device count 0") +
                        "); this engine has no CPU path.");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "Invalid HIP device ordinal.");

    rpf_engine* e = new rpf_engine();
    e->N = cfg->N;
    e->has_window = cfg->window != nullptr;
    e->n_buffers = cfg->n_buffers;
    e->buffer_capacity = static_cast<size_t>(cfg->buffer_capacity);
    e->device = cfg->device;
    e->flags = cfg->flags;
    e->use_dma = !(cfg->flags & RPF_FLAG_NO_LDS_DMA);
    e->variant = variant;
    e->fourstep = fourstep;
    e->mixed = mixed;
    e->bluestein = bluestein;
    e->bigblu = bigblu;
    e->generic = generic;
    e->queue_histogram.assign(e->n_buffers + 1, 0);
    e->pwr.assign(e->N, 0.0);

    auto cleanup = [&](int rc) {
        release_device(e);
        delete e;
        return rc;
    };
#define CREATE_TRY(call)                                                                 \
    do {                                                                                 \
        hipError_t err__ = (call);                                                       \
        if (err__ != hipSuccess) {                                                       \
            (void)hipGetLastError();     /* not the caller's next HIP call's problem */  \
            return cleanup(fail(nullptr, RPF_ERR_HARDWARE,                               \
                                std::string(#call) + ": " + hipGetErrorString(err__)));  \
        }                                                                                \
    } while (0)

    DeviceScope on_device(e->device);       // the caller's current device is restored on return
    CREATE_TRY(on_device.status());
    for (hipStream_t& cs : e->copy_streams) CREATE_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&e->compute_stream, hipStreamNonBlocking));

    // "plan": twiddle table on the device (where fftwf_plan_dft_1d stands, datastore.cxx:32)
    std::vector<rpf::cf> tw;
    int blu_m1 = 0, blu_m2 = 0;
    if (e->bigblu) rpf::bigblu_lengths(e->N, &e->blu_M, &blu_m1, &blu_m2);
    if (!e->generic) {
        rpf::make_twiddles(e->bluestein ? rpf::bluestein_length(e->N) : e->bigblu ? e->blu_M : e->N, tw);
        CREATE_TRY(hipMalloc(&e->d_twiddles, sizeof(rpf::cf) * tw.size()));
        CREATE_TRY(hipMemcpy(e->d_twiddles, tw.data(), sizeof(rpf::cf) * tw.size(), hipMemcpyHostToDevice));
    }
    if (e->has_window) {
        CREATE_TRY(hipMalloc(&e->d_window, sizeof(float) * e->N));
        CREATE_TRY(hipMemcpy(e->d_window, cfg->window, sizeof(float) * e->N, hipMemcpyHostToDevice));
    }
    size_t partial_slots = 0, partial_len = e->N;
    std::vector<float> blu_g, blu_bhat;
    const bool gen_blu = e->generic && rpf::generic_length(e->N) != e->N;
    if (e->bluestein || e->bigblu || gen_blu) {
        std::vector<float>&g = blu_g, &bhat = blu_bhat;
        rpf::make_bluestein_tables(e->N, cfg->window, g, bhat);
        CREATE_TRY(hipMalloc(&e->d_chirp, sizeof(float) * g.size()));
        CREATE_TRY(hipMemcpy(e->d_chirp, g.data(), sizeof(float) * g.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&e->d_bhat, sizeof(float) * bhat.size()));
        CREATE_TRY(hipMemcpy(e->d_bhat, bhat.data(), sizeof(float) * bhat.size(), hipMemcpyHostToDevice));
    }
    if (e->generic) {
        std::vector<rpf::cf> t0, t1;
        rpf::generic_twiddle_tables(e->N, t0, t1, &e->gen_h);
        CREATE_TRY(hipMalloc(&e->d_tw_sub, sizeof(rpf::cf) * t0.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub, t0.data(), sizeof(rpf::cf) * t0.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&e->d_tw_sub2, sizeof(rpf::cf) * t1.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub2, t1.data(), sizeof(rpf::cf) * t1.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&e->d_scratch, rpf::generic_scratch_bytes(e->N)));
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, e->device));
        e->plan.grid = prop.multiProcessorCount;
        e->plan.block = 256;
        e->plan.fpw = rpf::generic_batch(e->N);
        e->plan.lds_bytes = 0;
        partial_slots = 1;
    } else if (e->mixed) {
        CREATE_TRY(rpf::plan_mixed(e->N, e->variant, e->has_window, e->device, &e->plan));
        partial_slots = e->plan.grid;
    } else if (e->bluestein) {
        CREATE_TRY(rpf::plan_bluestein(e->N, e->device, &e->plan));
        partial_slots = e->plan.grid;
    } else if (e->bigblu) {
        CREATE_TRY(rpf::bigblu_prepare(e->N, e->device, &e->plan));
        std::vector<rpf::cf> tws;
        rpf::make_twiddles(blu_m1, tws);
        CREATE_TRY(hipMalloc(&e->d_tw_sub, sizeof(rpf::cf) * tws.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub, tws.data(), sizeof(rpf::cf) * tws.size(), hipMemcpyHostToDevice));
        rpf::make_twiddles(blu_m2, tws);
        CREATE_TRY(hipMalloc(&e->d_tw_sub2, sizeof(rpf::cf) * tws.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub2, tws.data(), sizeof(rpf::cf) * tws.size(), hipMemcpyHostToDevice));
        // (the intermediate starts at 256 MB and grows with the launches: ensure_scratch)
        e->scratch_per_frame = rpf::bigblu_scratch_bytes_per_frame(e->N);
        e->scratch_max = rpf::bigblu_scratch_bytes(e->N);
        e->scratch_bytes = std::min<size_t>(e->scratch_max, std::max<size_t>(e->scratch_per_frame, static_cast<size_t>(256) << 20));
        CREATE_TRY(hipMalloc(&e->d_scratch, e->scratch_bytes));
        partial_slots = rpf::bigblu_partial_slots(e->N);
        partial_len = e->blu_M;
        // the kernels read chirp, kernel spectrum and inter-step twiddles in their own lane order
        std::vector<rpf::cf> g_t, bhat_t, step_tw, step_tw2;
        rpf::bigblu_tables(e->N, reinterpret_cast<const rpf::cf*>(blu_g.data()),
                           reinterpret_cast<const rpf::cf*>(blu_bhat.data()), g_t, bhat_t, step_tw, step_tw2);
        (void)hipFree(e->d_chirp);
        e->d_chirp = nullptr;
        CREATE_TRY(hipMalloc(&e->d_chirp, sizeof(rpf::cf) * g_t.size()));
        CREATE_TRY(hipMemcpy(e->d_chirp, g_t.data(), sizeof(rpf::cf) * g_t.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(e->d_bhat, bhat_t.data(), sizeof(rpf::cf) * bhat_t.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(e->d_twiddles, step_tw.data(), sizeof(rpf::cf) * step_tw.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&e->d_step2, sizeof(rpf::cf) * step_tw2.size()));
        CREATE_TRY(hipMemcpy(e->d_step2, step_tw2.data(), sizeof(rpf::cf) * step_tw2.size(), hipMemcpyHostToDevice));
    } else if (e->fourstep) {
        CREATE_TRY(rpf::fourstep_prepare(e->N, e->device, &e->plan));
        int n1 = 0, n2 = 0;
        rpf::fourstep_sub_lengths(e->N, &n1, &n2);
        std::vector<rpf::cf> tws;
        rpf::make_twiddles(n1, tws);
        CREATE_TRY(hipMalloc(&e->d_tw_sub, sizeof(rpf::cf) * tws.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub, tws.data(), sizeof(rpf::cf) * tws.size(), hipMemcpyHostToDevice));
        rpf::make_twiddles(n2, tws);
        CREATE_TRY(hipMalloc(&e->d_tw_sub2, sizeof(rpf::cf) * tws.size()));
        CREATE_TRY(hipMemcpy(e->d_tw_sub2, tws.data(), sizeof(rpf::cf) * tws.size(), hipMemcpyHostToDevice));
        // The fused kernel (one persistent launch, the intermediate stays in each XCD's L2) where
        // the device is the 8 x 32-CU part it is written for and its teams assemble; else K2a/K2b.
        int fused_grid = 0;
        if (!(cfg->flags & RPF_FLAG_NO_FOURSTEP_FUSED) &&
            rpf::fourstep_fused_prepare(e->N, e->device, &fused_grid) == hipSuccess) {
            CREATE_TRY(hipMalloc(&e->d_fused_scratch, rpf::fourstep_fused_scratch_bytes(e->N)));
            CREATE_TRY(hipMalloc(&e->d_fused_ctl, rpf::fourstep_fused_ctl_bytes()));
            CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_fused_words), 64, hipHostMallocMapped));
            std::memset(e->h_fused_words, 0, 64);
            CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_fused_words), e->h_fused_words, 0));
            e->fused = true;
            // (room for either path's partial spectra: a fused launch that gives up is re-run on K2a/K2b)
            partial_slots = std::max<size_t>(rpf::fourstep_fused_slots(e->N), rpf::fourstep_partial_slots(e->N));
        } else {
            (void)hipGetLastError();
            e->scratch_per_frame = rpf::fourstep_scratch_bytes_per_frame(e->N);
            e->scratch_max = rpf::fourstep_scratch_bytes(e->N);
            e->scratch_bytes = std::min<size_t>(e->scratch_max, static_cast<size_t>(256) << 20);
            CREATE_TRY(hipMalloc(&e->d_scratch, e->scratch_bytes));
            partial_slots = rpf::fourstep_partial_slots(e->N);
        }
        // K2a reads the inter-step twiddles and the window in its own lane order
        std::vector<rpf::cf> step_tw;
        std::vector<float> window_t;
        rpf::fourstep_tables(e->N, cfg->window, step_tw, window_t);
        CREATE_TRY(hipMemcpy(e->d_twiddles, step_tw.data(), sizeof(rpf::cf) * step_tw.size(), hipMemcpyHostToDevice));
        if (e->has_window)
            CREATE_TRY(hipMemcpy(e->d_window, window_t.data(), sizeof(float) * window_t.size(), hipMemcpyHostToDevice));
    } else {
        CREATE_TRY(rpf::plan_launch(e->N, e->variant, e->has_window, true, e->device, &e->plan));
        rpf::LaunchInfo tmp;
        CREATE_TRY(rpf::plan_launch(e->N, e->variant, e->has_window, false, e->device, &tmp));
        e->plan.grid = std::min(e->plan.grid, tmp.grid);
        partial_slots = e->plan.grid + rpf::kMaxHops;    // a workgroup leaves one partial per hop it touches
    }
    CREATE_TRY(hipMalloc(&e->d_partial, sizeof(double) * partial_len * partial_slots));
    CREATE_TRY(hipMalloc(&e->d_pwr, sizeof(double) * e->N));
    CREATE_TRY(hipMemset(e->d_pwr, 0, sizeof(double) * e->N));
    if (e->fused) {
        // Prove once that the eight teams assemble on this device (one round per team on a dummy
        // stream); if they do not, this engine uses the two-kernel path.
        const size_t round_bytes = 2u * 262144u * 8u;         // FR frames of N samples for each of 8 teams
        void* d_dummy = nullptr;
        struct FreeOnExit {                                   // (every exit of this block, CREATE_TRY's included)
            void*& p;
            ~FreeOnExit() { if (p) (void)hipFree(p); }
        } free_dummy{d_dummy};
        CREATE_TRY(hipMalloc(&d_dummy, round_bytes));
        CREATE_TRY(hipMemset(d_dummy, 0x80, round_bytes));
        bool aborted = true;
        hipError_t lerr = rpf::launch_fourstep_fused(e->N, e->has_window, true, static_cast<const uint8_t*>(d_dummy),
                                                     static_cast<long>(round_bytes / (2u * static_cast<size_t>(e->N))),
                                                     e->d_tw_sub, e->d_tw_sub2, e->d_twiddles, e->d_window, e->d_fused_scratch,
                                                     e->d_partial, e->d_fused_ctl, e->compute_stream);
        if (lerr == hipSuccess) lerr = rpf::fourstep_fused_aborted(e->d_fused_ctl, e->compute_stream, &aborted);
        if (lerr != hipSuccess || aborted) {
            (void)hipGetLastError();
            retire_fused(e);
            (void)hipFree(e->d_fused_scratch);          // (nothing is in flight here)
            e->d_fused_scratch = nullptr;
            e->scratch_bytes = std::min<size_t>(e->scratch_max, static_cast<size_t>(256) << 20);
            CREATE_TRY(hipMalloc(&e->d_scratch, e->scratch_bytes));
        }
    }

    // buffer pool: pinned host memory (datastore.cxx:27-28)
    // ONE allocation, the buffers side by side: what the producer fills in order the consumer can send in one copy.
    // (If that much pinned memory is not to be had in one piece: buffer by buffer, as before.)
    e->pool.resize(e->n_buffers);
    {
        void* p = nullptr;
        if (hipHostMalloc(&p, e->buffer_capacity * static_cast<size_t>(e->n_buffers), hipHostMallocDefault) == hipSuccess)
            e->pool_base = static_cast<uint8_t*>(p);
        else
            (void)hipGetLastError();
    }
    for (size_t k = 0; k < e->pool.size(); ++k) {
        HostBuffer& b = e->pool[k];
        if (e->pool_base) {
            b.data = e->pool_base + k * e->buffer_capacity;
        } else {
            void* p = nullptr;
            CREATE_TRY(hipHostMalloc(&p, e->buffer_capacity, hipHostMallocDefault));
            b.data = static_cast<uint8_t*>(p);
        }
        b.size = e->buffer_capacity;
        CREATE_TRY(hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
        e->empty_buffers.push_back(&b);
    }
    // device staging ring: head room for a carried partial frame + one buffer
    e->head_room = ((2 * static_cast<size_t>(e->N)) + 255) / 256 * 256;
    // a slot (= one transform launch) holds as many buffers as fit 32 MB, at least one -- more than the pool has where the
    // buffers are small: they return to the producer when their copy lands, not when the slot is launched
    e->coalesce = std::max<size_t>(1, (32u << 20) / e->buffer_capacity);
    e->staging.resize(kStagingSlots);
    for (auto& s : e->staging) {
        void* p = nullptr;
        CREATE_TRY(hipMalloc(&p, e->head_room + e->coalesce * e->buffer_capacity));
        s.base = static_cast<uint8_t*>(p);
        CREATE_TRY(hipEventCreateWithFlags(&s.kernel_done, hipEventDisableTiming));
        for (hipEvent_t& ev : s.copy_done) CREATE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    static_assert(kStagingSlots <= 3, "h_fused_words[0 .. 2]: one verdict word per staging slot");
    if (e->h_fused_words)
        for (size_t k = 0; k < e->staging.size(); ++k) e->staging[k].verdict = e->h_fused_words + k;
#undef CREATE_TRY
    *out = e;
    return RPF_OK;
}

void rpf_engine_destroy(rpf_engine* e)
{
    if (!e) return;
    if (e->worker_running) {
        int64_t dummy;
        (void)rpf_finish(e, &dummy);
    }
    {
        DeviceScope on_device(e->device);
        release_device(e);
    }
    delete e;
}

const char* rpf_last_error(const rpf_engine* e) { return e ? e->last_error.c_str() : g_last_error.c_str(); }

int rpf_begin(rpf_engine* e, int64_t repeats)
{
    if (!e) return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_begin: NULL engine");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_begin: acquisition already running");
    if (repeats < 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "Argument to 'repeats' must be a positive number.");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    // acquisition.cxx:252-254
    std::fill(e->pwr.begin(), e->pwr.end(), 0.0);
    HIP_TRY(e, hipMemsetAsync(e->d_pwr, 0, sizeof(double) * e->N, e->compute_stream));
    {
        std::lock_guard<std::mutex> lock(e->status_mutex);
        e->acquisition_finished = false;
    }
    e->repeats_done = 0;
    e->repeats = repeats;
    e->worker_rc = RPF_OK;
    e->worker_error.clear();
    // acquisition.cxx:256
    try {
        e->worker = std::thread(worker_main, e);
    } catch (const std::exception& ex) {      // (no exception crosses the C boundary)
        return fail(e, RPF_ERR_HARDWARE, std::string("rpf_begin: cannot start the consumer thread: ") + ex.what());
    }
    e->worker_running = true;
    return RPF_OK;
}

int rpf_buffer_acquire(rpf_engine* e, uint8_t** buf, size_t* capacity)
{
    if (!e || !buf) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_acquire: NULL argument");
    // acquisition.cxx:278-285
    std::unique_lock<std::mutex> lock(e->status_mutex);
    e->queue_histogram[e->empty_buffers.size()]++;
    wait_briefly_then_block(lock, e->status_change, [&]() { return !e->empty_buffers.empty(); });
    HostBuffer* b = e->empty_buffers.front();
    e->empty_buffers.pop_front();
    lock.unlock();
    *buf = b->data;
    if (capacity) *capacity = e->buffer_capacity;
    return RPF_OK;
}

static HostBuffer* find_buffer(rpf_engine* e, const uint8_t* p)
{
    for (auto& b : e->pool)
        if (b.data == p) return &b;
    return nullptr;
}

int rpf_buffer_submit(rpf_engine* e, uint8_t* buf, size_t nbytes)
{
    if (!e) return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_submit: NULL engine");
    HostBuffer* b = find_buffer(e, buf);
    if (!b) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_submit: not an engine buffer");
    if (nbytes > e->buffer_capacity || (nbytes % 2) != 0)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_submit: size must be even and <= capacity");
    if (!e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_submit: no acquisition running");
    b->size = nbytes;   // buffer.resize(dataNeeded), acquisition.cxx:302
    // acquisition.cxx:320-323
    std::lock_guard<std::mutex> lock(e->status_mutex);
    e->occupied_buffers.push_back(b);
    e->status_change.notify_all();
    return RPF_OK;
}

int rpf_buffer_unget(rpf_engine* e, uint8_t* buf)
{
    if (!e) return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_unget: NULL engine");
    HostBuffer* b = find_buffer(e, buf);
    if (!b) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_buffer_unget: not an engine buffer");
    // acquisition.cxx:310-314 (no notify needed)
    std::lock_guard<std::mutex> lock(e->status_mutex);
    e->empty_buffers.push_front(b);
    return RPF_OK;
}

int rpf_finish(rpf_engine* e, int64_t* repeats_done)
{
    if (!e) return fail(nullptr, RPF_ERR_INVALID_ARGUMENT, "rpf_finish: NULL engine");
    if (!e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_finish: no acquisition running");
    // acquisition.cxx:343-347
    {
        std::lock_guard<std::mutex> lock(e->status_mutex);
        e->acquisition_finished = true;
        e->status_change.notify_all();
    }
    e->worker.join();
    e->worker_running = false;
    if (repeats_done) *repeats_done = e->repeats_done;
    if (e->worker_rc != RPF_OK) return fail(e, e->worker_rc, e->worker_error);
    return RPF_OK;
}

int rpf_get_power(const rpf_engine* e, double* out)
{
    if (!e || !out) return RPF_ERR_INVALID_ARGUMENT;
    // datastore.h:40-47: results are read only after the worker has been joined
    if (e->worker_running)
        return fail(const_cast<rpf_engine*>(e), RPF_ERR_INVALID_ARGUMENT,
                    "rpf_get_power: acquisition still running (call rpf_finish first)");
    std::memcpy(out, e->pwr.data(), sizeof(double) * e->N);
    return RPF_OK;
}

int rpf_copy_power_device(const rpf_engine* e, double* d_dst, void* hip_stream, int dst_device)
{
    if (!e || !d_dst) return RPF_ERR_INVALID_ARGUMENT;
    rpf_engine* me = const_cast<rpf_engine*>(e);
    if (e->worker_running)
        return fail(me, RPF_ERR_INVALID_ARGUMENT, "rpf_copy_power_device: acquisition still running (call rpf_finish first)");
    DeviceScope on_device(e->device);
    HIP_TRY(me, on_device.status());
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (dst_device == e->device)
        HIP_TRY(me, hipMemcpyAsync(d_dst, e->d_pwr, sizeof(double) * e->N, hipMemcpyDeviceToDevice, s));
    else
        HIP_TRY(me, hipMemcpyPeerAsync(d_dst, dst_device, e->d_pwr, e->device, sizeof(double) * e->N, s));
    HIP_TRY(me, hipStreamSynchronize(s));
    return RPF_OK;
}

int64_t rpf_get_repeats_done(const rpf_engine* e) { return e ? e->repeats_done : 0; }

int rpf_get_histogram(const rpf_engine* e, int* out)
{
    if (!e || !out) return RPF_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(const_cast<rpf_engine*>(e)->status_mutex);
    std::memcpy(out, e->queue_histogram.data(), sizeof(int) * e->queue_histogram.size());
    return RPF_OK;
}

// The producer side of rpf_accumulate: pageable stream -> pinned hand-off buffer.  One thread copies
// ~10 GB/s, a fifth of what the H2D link behind it takes, so a large buffer is filled by several threads
// (measured with 5 x 105 MB buffers: 8.0 -> see bench.py's end_to_end; small buffers are not worth a fork).
static void fill_buffer(uint8_t* dst, const uint8_t* src, size_t n)
{
    constexpr size_t kPiece = 4u << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t want = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n / kPiece);
    if (want < 2) {
        std::memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> helpers;
    const size_t share = (n / want + 63) & ~static_cast<size_t>(63);
    size_t forked_from = n;          // [0, forked_from) is copied by this thread: its own share, plus what no helper took
    try {
        helpers.reserve(want - 1);
        for (size_t t = want - 1; t >= 1; --t) {      // from the far end, so that what is left over is one run
            const size_t lo = t * share, hi = std::min(n, lo + share);
            if (lo < hi) {
                helpers.emplace_back([=]() { std::memcpy(dst + lo, src + lo, hi - lo); });
                forked_from = lo;
            }
        }
    } catch (...) {
        // std::thread could not start (thread limit, out of memory): nothing may leave rpf_accumulate's
        // extern "C" frame as an exception -- this thread copies the rest itself
    }
    std::memcpy(dst, src, std::min(n, forked_from));
    for (std::thread& h : helpers) h.join();
}

int rpf_accumulate(rpf_engine* e, const uint8_t* stream, size_t nbytes, int64_t repeats,
                   double* pwr_out, int64_t* repeats_done)
{
    if (!e || (!stream && nbytes)) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate: NULL argument");
    int rc = rpf_begin(e, repeats);
    if (rc != RPF_OK) return rc;
    size_t pos = 0;
    const size_t cap = e->buffer_capacity;
    // A stream the caller has pinned (rpf_stream_register): no copy into the pool -- its pieces go to the consumer as they
    // lie, and the consumer sends neighbours in one copy, a staging slot at a time.
    std::vector<HostBuffer> pieces;
    for (const auto& r : e->registered) {
        if (stream >= r.first && stream + nbytes <= r.first + r.second) {
            const size_t piece = std::min<size_t>(e->coalesce * e->buffer_capacity, static_cast<size_t>(8) << 20) & ~static_cast<size_t>(1);
            const size_t even = nbytes & ~static_cast<size_t>(1);
            pieces.reserve(even / piece + 1);
            for (size_t at = 0; at < even; at += piece) {
                HostBuffer hb;
                hb.data = const_cast<uint8_t*>(stream) + at;
                hb.size = std::min(piece, even - at);
                hb.external = true;
                pieces.push_back(hb);
            }
            {
                std::lock_guard<std::mutex> lock(e->status_mutex);
                for (HostBuffer& hb : pieces) e->occupied_buffers.push_back(&hb);
                e->status_change.notify_all();
            }
            pos = nbytes;
            break;
        }
    }
    while (pos < nbytes) {
        uint8_t* buf = nullptr;
        rc = rpf_buffer_acquire(e, &buf, nullptr);
        if (rc != RPF_OK) break;
        const size_t n = std::min(cap, (nbytes - pos) & ~static_cast<size_t>(1));
        if (n == 0) {
            rpf_buffer_unget(e, buf);
            break;
        }
        fill_buffer(buf, stream + pos, n);
        rc = rpf_buffer_submit(e, buf, n);
        if (rc != RPF_OK) break;
        pos += n;
    }
    int64_t done = 0;
    int rc2 = rpf_finish(e, &done);
    if (rc == RPF_OK) rc = rc2;
    if (repeats_done) *repeats_done = done;
    if (rc == RPF_OK && pwr_out) std::memcpy(pwr_out, e->pwr.data(), sizeof(double) * e->N);
    return rc;
}

int rpf_accumulate_device(rpf_engine* e, const void* d_stream, size_t nbytes, int64_t repeats,
                          double* d_pwr_out, void* hip_stream, int64_t* repeats_done)
{
    if (!e || !d_pwr_out || (!d_stream && nbytes))
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device: NULL argument");
    if (e->worker_running)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device: acquisition running");
    if (repeats < 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "Argument to 'repeats' must be a positive number.");
    if (reinterpret_cast<uintptr_t>(d_stream) & 1)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device: d_stream must be at least 2-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_pwr_out) & 15)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device: d_pwr_out must be 16-byte aligned");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    hipStream_t s = static_cast<hipStream_t>(hip_stream);   // NULL = the HIP null stream
    int64_t nframes = static_cast<int64_t>(nbytes / (2 * static_cast<size_t>(e->N)));
    nframes = std::min(nframes, repeats);
    if (repeats_done) *repeats_done = nframes;
    if (nframes == 0) {
        HIP_TRY(e, hipMemsetAsync(d_pwr_out, 0, sizeof(double) * e->N, s));
        return RPF_OK;
    }
    return launch_frames(e, static_cast<const uint8_t*>(d_stream), nframes, d_pwr_out,
                         /*accumulate=*/false, s);
}

int rpf_device_fused(rpf_engine* e, const void* d_stream, size_t nbytes, int64_t repeats,
                     void* hip_stream, int64_t* repeats_done)
{
    if (!e || !d_stream) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_fused: NULL argument");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_fused: acquisition running");
    if (reinterpret_cast<uintptr_t>(d_stream) & 1)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_fused: d_stream must be at least 2-byte aligned");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    int64_t nframes = static_cast<int64_t>(nbytes / (2 * static_cast<size_t>(e->N)));
    nframes = std::min(nframes, repeats);
    if (nframes < 1) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_fused: no whole frame");
    if (repeats_done) *repeats_done = nframes;
    if (e->fused) note_device_path_aborts(e);
    int nslots = 0;
    int rc = launch_transform(e, static_cast<const uint8_t*>(d_stream), nframes,
                              static_cast<hipStream_t>(hip_stream), &nslots);
    if (rc == RPF_OK) {
        e->last_slots = nslots;
        e->last_hops = 0;
    }
    return rc;
}

int rpf_device_reduce(rpf_engine* e, double* d_pwr_out, void* hip_stream)
{
    if (!e || !d_pwr_out) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_reduce: NULL argument");
    if (reinterpret_cast<uintptr_t>(d_pwr_out) & 15)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_reduce: d_pwr_out must be 16-byte aligned");
    if (e->last_slots < 1 && e->last_hops < 1) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_device_reduce: nothing to reduce");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    if (e->last_hops > 0) {
        HIP_TRY(e, rpf::launch_reduce_hops(e->d_partial, e->last_ranges, e->last_hops, e->N, d_pwr_out,
                                           /*accumulate=*/false, static_cast<hipStream_t>(hip_stream), e->plan.partial_f32));
        return RPF_OK;
    }
    HIP_TRY(e, rpf::launch_reduce(e->d_partial, e->last_slots, e->N, d_pwr_out,
                                  /*accumulate=*/false, static_cast<hipStream_t>(hip_stream),
                                  e->plan.partial_f32, e->bigblu ? static_cast<size_t>(e->blu_M) : 0));
    if (e->last_was_fused)
        HIP_TRY(e, rpf::launch_fused_verdict(e->d_fused_ctl, d_pwr_out, e->N, e->d_fused_words + 3, e->d_fused_words + 4,
                                             static_cast<hipStream_t>(hip_stream)));
    return RPF_OK;
}

// Shared argument checks of the hop entries; fills frames[h] = min(repeats[h], nbytes[h] / 2N).
static int check_hops(rpf_engine* e, const char* who, const void* const* d_streams, const size_t* nbytes,
                      const int64_t* repeats, int H, std::vector<int64_t>* frames)
{
    if (!e || !d_streams || !nbytes || !repeats || H < 1)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL argument or no hop");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, std::string(who) + ": acquisition running");
    frames->resize(H);
    for (int h = 0; h < H; ++h) {
        if (repeats[h] < 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "Argument to 'repeats' must be a positive number.");
        if (!d_streams[h] && nbytes[h]) return fail(e, RPF_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL stream");
        if (reinterpret_cast<uintptr_t>(d_streams[h]) & 1)
            return fail(e, RPF_ERR_INVALID_ARGUMENT, std::string(who) + ": streams must be at least 2-byte aligned");
        (*frames)[h] = std::min<int64_t>(static_cast<int64_t>(nbytes[h] / (2 * static_cast<size_t>(e->N))), repeats[h]);
    }
    return RPF_OK;
}

static bool is_k1(const rpf_engine* e)
{
    return !(e->fourstep || e->mixed || e->bluestein || e->bigblu || e->generic);
}

int rpf_max_hops_per_launch(void) { return rpf::kMaxHops; }

int rpf_accumulate_device_hops(rpf_engine* e, const void* const* d_streams, const size_t* nbytes,
                               const int64_t* repeats, int n_hops, double* d_pwr_out, void* hip_stream,
                               int64_t* repeats_done)
{
    std::vector<int64_t> frames;
    int rc = check_hops(e, "rpf_accumulate_device_hops", d_streams, nbytes, repeats, n_hops, &frames);
    if (rc != RPF_OK) return rc;
    if (!d_pwr_out) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device_hops: NULL argument");
    if (reinterpret_cast<uintptr_t>(d_pwr_out) & 15)
        return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_accumulate_device_hops: d_pwr_out must be 16-byte aligned");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (repeats_done)
        for (int h = 0; h < n_hops; ++h) repeats_done[h] = frames[h];
    const size_t N = static_cast<size_t>(e->N);
    if (!is_k1(e)) {
        // the other kernel families run one acquisition per launch set
        for (int h = 0; h < n_hops; ++h) {
            if (frames[h] == 0) {
                HIP_TRY(e, hipMemsetAsync(d_pwr_out + h * N, 0, sizeof(double) * N, s));
                continue;
            }
            rc = launch_frames(e, static_cast<const uint8_t*>(d_streams[h]), frames[h], d_pwr_out + h * N,
                               /*accumulate=*/false, s);
            if (rc != RPF_OK) return rc;
        }
        return RPF_OK;
    }
    // K1: ONE persistent launch + ONE reduce per rpf::kMaxHops hops
    for (int h0 = 0; h0 < n_hops; h0 += rpf::kMaxHops) {
        const int hc = std::min(rpf::kMaxHops, n_hops - h0);
        const uint8_t* ptrs[rpf::kMaxHops];
        for (int h = 0; h < hc; ++h) ptrs[h] = static_cast<const uint8_t*>(d_streams[h0 + h]);
        rpf::SlotRanges slots;
        int nslots = 0;
        rc = launch_fused_hops(e, ptrs, frames.data() + h0, hc, s, &slots, &nslots);
        if (rc != RPF_OK) return rc;
        // (a hop without a whole frame has an empty slot range: the reduce writes zeros)
        HIP_TRY(e, rpf::launch_reduce_hops(e->d_partial, slots, hc, e->N, d_pwr_out + h0 * N, /*accumulate=*/false, s,
                                           e->plan.partial_f32));
        e->last_slots = nslots;
        e->last_ranges = slots;
        e->last_hops = hc;
    }
    return RPF_OK;
}

int rpf_device_fused_hops(rpf_engine* e, const void* const* d_streams, const size_t* nbytes,
                          const int64_t* repeats, int n_hops, void* hip_stream, int64_t* repeats_done)
{
    std::vector<int64_t> frames;
    int rc = check_hops(e, "rpf_device_fused_hops", d_streams, nbytes, repeats, n_hops, &frames);
    if (rc != RPF_OK) return rc;
    if (!is_k1(e) || n_hops > rpf::kMaxHops)
        return fail(e, RPF_ERR_INVALID_ARGUMENT,
                    "rpf_device_fused_hops: needs a size the LDS-resident kernel serves and at most rpf_max_hops_per_launch() hops");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    if (repeats_done)
        for (int h = 0; h < n_hops; ++h) repeats_done[h] = frames[h];
    const uint8_t* ptrs[rpf::kMaxHops];
    for (int h = 0; h < n_hops; ++h) ptrs[h] = static_cast<const uint8_t*>(d_streams[h]);
    int nslots = 0;
    rc = launch_fused_hops(e, ptrs, frames.data(), n_hops, static_cast<hipStream_t>(hip_stream), &e->last_ranges, &nslots);
    if (rc != RPF_OK) return rc;
    e->last_slots = nslots;
    e->last_hops = n_hops;
    return RPF_OK;
}

int rpf_stream_register(rpf_engine* e, const void* stream, size_t nbytes)
{
    if (!e || !stream || nbytes == 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_stream_register: NULL or empty stream");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_stream_register: acquisition running");
    DeviceScope on_device(e->device);
    HIP_TRY(e, on_device.status());
    const hipError_t err = hipHostRegister(const_cast<void*>(stream), nbytes, hipHostRegisterDefault);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, RPF_ERR_HARDWARE, std::string("hipHostRegister: ") + hipGetErrorString(err));
    }
    e->registered.emplace_back(static_cast<const uint8_t*>(stream), nbytes);
    return RPF_OK;
}

int rpf_stream_unregister(rpf_engine* e, const void* stream)
{
    if (!e || !stream) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_stream_unregister: NULL argument");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_stream_unregister: acquisition running");
    for (size_t k = 0; k < e->registered.size(); ++k) {
        if (e->registered[k].first != stream) continue;
        DeviceScope on_device(e->device);
        HIP_TRY(e, on_device.status());
        HIP_TRY(e, hipHostUnregister(const_cast<void*>(stream)));
        e->registered.erase(e->registered.begin() + static_cast<long>(k));
        return RPF_OK;
    }
    return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_stream_unregister: not a registered stream");
}

int rpf_fused_status(rpf_engine* e, int* active, int64_t* launches_gave_up, int64_t* launches_recovered)
{
    if (!e) return RPF_ERR_INVALID_ARGUMENT;
    if (e->fused && !e->worker_running) note_device_path_aborts(e);
    if (active) *active = (e->fourstep && e->fused) ? 1 : 0;
    if (launches_gave_up) *launches_gave_up = e->h_fused_words ? __atomic_load_n(&e->h_fused_words[4], __ATOMIC_ACQUIRE) : 0;
    if (launches_recovered) *launches_recovered = e->fused_recovered;
    return RPF_OK;
}

int rpf_debug_fused_fault(rpf_engine* e, int mode, int skip, int count)
{
    if (!e || mode < 0 || mode > 2 || skip < 0) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_debug_fused_fault: bad argument");
    if (e->worker_running) return fail(e, RPF_ERR_INVALID_ARGUMENT, "rpf_debug_fused_fault: acquisition running");
    e->fused_fault_mode = mode;
    e->fused_fault_skip = skip;
    e->fused_fault_count = mode ? count : 0;
    return RPF_OK;
}

int rpf_last_launch_info(const rpf_engine* e, int* grid, int* block, int* frames_per_wg,
                         int* lds_bytes)
{
    if (!e) return RPF_ERR_INVALID_ARGUMENT;
    if (grid) *grid = e->last.grid ? e->last.grid : e->plan.grid;
    if (block) *block = e->plan.block;
    if (frames_per_wg) *frames_per_wg = e->plan.fpw;
    if (lds_bytes) *lds_bytes = e->plan.lds_bytes;
    return RPF_OK;
}

}  // extern "C"
