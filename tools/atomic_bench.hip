// How expensive is it to flush per-workgroup spectra (768 x 4096 doubles) with
// f64 atomics instead of 25 MB of partial stores?  Compares agent-scope and
// workgroup-scope (L2-local) atomics, one copy vs one copy per XCD, and checks
// that HW_REG_XCC_ID really names the XCD the workgroup runs on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int N = 4096;

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7; }

template <int MODE>   // 0 agent/1 copy, 1 agent/8 copies, 2 workgroup-scope/8 copies, 3 plain partial stores
__global__ __launch_bounds__(256) void flush(double* acc, int* xcc_seen)
{
    const int x = xcc_id();
    if (threadIdx.x == 0 && xcc_seen) xcc_seen[blockIdx.x] = x;
    double* dst = acc + (MODE == 0 ? 0 : MODE == 3 ? (size_t)blockIdx.x * N : (size_t)x * N);
#pragma unroll
    for (int i = 0; i < N / 256; ++i) {
        const int bin = i * 256 + threadIdx.x;
        const double v = 1.0 + bin * 1e-3;
        if (MODE == 3) dst[bin] = v;
        else if (MODE == 2) __hip_atomic_fetch_add(dst + bin, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(dst + bin, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
int run(const char* name, double* d_acc, int* d_xcc, int grid)
{
    const size_t bytes = (MODE == 3 ? (size_t)grid : 8) * N * sizeof(double);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipMemset(d_acc, 0, bytes));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(flush<MODE>, dim3(grid), dim3(256), 0, 0, d_acc, d_xcc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<double> h((MODE == 3 ? (size_t)grid : 8) * N);
    CHECK(hipMemcpy(h.data(), d_acc, bytes, hipMemcpyDeviceToHost));
    // verify the total per bin
    int bad = 0;
    const size_t copies = (MODE == 3 ? (size_t)grid : 8);
    for (int bin = 0; bin < N; ++bin) {
        double s = 0;
        for (size_t c = 0; c < copies; ++c) s += h[c * N + bin];
        const double want = grid * (1.0 + bin * 1e-3);
        if (fabs(s - want) > 1e-9 * want) ++bad;
    }
    printf("%-34s grid=%d  %.2f us  bad_bins=%d\n", name, grid, best * 1e3, bad);
    return 0;
}

int main()
{
    const int grid = 768;
    double* d_acc; int* d_xcc;
    CHECK(hipMalloc(&d_acc, (size_t)grid * N * sizeof(double)));
    CHECK(hipMalloc(&d_xcc, grid * sizeof(int)));
    run<3>("partial stores (25 MB)", d_acc, d_xcc, grid);
    run<0>("agent atomics, 1 copy", d_acc, d_xcc, grid);
    run<1>("agent atomics, copy per XCD", d_acc, d_xcc, grid);
    run<2>("workgroup-scope atomics, per XCD", d_acc, d_xcc, grid);
    std::vector<int> x(grid);
    CHECK(hipMemcpy(x.data(), d_xcc, grid * sizeof(int), hipMemcpyDeviceToHost));
    int match = 0, hist[8] = {0};
    for (int b = 0; b < grid; ++b) { match += (x[b] == b % 8); hist[x[b] & 7]++; }
    printf("xcc_id == blockIdx %% 8 for %d of %d blocks; per-XCD counts:", match, grid);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("\n");
    return 0;
}
