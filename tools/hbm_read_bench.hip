// Read-only streaming kernel: the measured HBM read rate SURVEY.md 8(d) asks to
// report beside the 8 TB/s spec figure.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bench.hip -o tools/hbm_read_bench && tools/hbm_read_bench
// Every lane reads 16 B per load, 8 independent loads in flight, grid-stride over
// a buffer larger than the 256 MB Infinity Cache (so the rate is HBM's), and the
// same over 82 MB (the size of one C2 launch: what a cache-resident replay sees).
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

int main()
{
    const size_t sizes[] = {static_cast<size_t>(2) << 30, 81920000};
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
        CHECK(hipFree(buf));
    }
    return 0;
}
