// Read-only streaming kernel: the measured HBM read rate SURVEY.md 8(d) asks to
// report beside the 8 TB/s spec figure.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bench.hip -o tools/hbm_read_bench && tools/hbm_read_bench
// Every lane reads 16 B per load, 8 independent loads in flight, grid-stride over
// a buffer larger than the 256 MB Infinity Cache (so the rate is HBM's), and the
// same over 82 MB (the size of one C2 launch: what a cache-resident replay sees).
// Round 3: the whole ladder of working-set sizes (does data that stays in the 256 MB Infinity Cache,
// or in the 8 x 4 MB L2s, stream any faster?) and a write-only pass of the same shapes -- the
// question behind the four-step kernels' intermediate.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, size_t n, unsigned* sink)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;   // never true for the fill pattern; keeps the loads alive
}

__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ p, size_t n, unsigned seed)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const uint4 v = {seed, seed + 1, seed + 2, static_cast<unsigned>(threadIdx.x)};
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

int main()
{
    const size_t sizes[] = {static_cast<size_t>(2) << 30, static_cast<size_t>(512) << 20, static_cast<size_t>(256) << 20,
                            static_cast<size_t>(128) << 20, 81920000, static_cast<size_t>(32) << 20,
                            static_cast<size_t>(16) << 20, static_cast<size_t>(4) << 20};
    unsigned* sink;
    CHECK(hipMalloc(&sink, 4));
    for (size_t bytes : sizes) {
        uint4* buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0x5a, bytes));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        const int grid = 256 * 8;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink);
        const int reps = 20;
        CHECK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("read-only stream of %zu B: %.3f ms per pass = %.0f GB/s\n", bytes, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, 1u);
        CHECK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, 2u + r);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("write-only stream of %zu B: %.3f ms per pass = %.0f GB/s\n", bytes, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
        // write then read back (the intermediate's round trip), back to back
        CHECK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, 50u + r);
            hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink);
        }
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("write + read back of %zu B: %.3f ms per round trip = %.0f GB/s of traffic\n", bytes, ms / reps, 2.0 * bytes / (ms / reps * 1e-3) / 1e9);
        CHECK(hipFree(buf));
    }
    return 0;
}
