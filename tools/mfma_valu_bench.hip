// Does the matrix pipe of gfx950 run beside vector arithmetic?  Per wave and loop iteration:
//   P = 128 v_pk_fma_f32 (16 independent chains)        -- the packed-f32 form K1 uses
//   S = 256 v_fma_f32 (32 independent chains)           -- the same flops, one lane-pair per instruction
//   D = 128 v_fma_f64 (16 chains)                       -- K1's double accumulate
//   M = 16 v_mfma_f32_32x32x16_f16 (two accumulators)
//   mode 0: P   1: S   2: D   3: M   4: P with M interleaved (1 per 8)   5: S with M interleaved (1 per 16)
//   6: D with M interleaved (1 per 8)   7: P then M   8: S then M
// Grid = 256 workgroups of 512 threads (two waves per SIMD, K1's shape).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_bench.hip -o tools/mfma_valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(i) & 15]) : "v"(b), "v"(c));
#define SC(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i) & 31]) : "v"(b.x), "v"(c.x));
#define DP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[(i) & 15]) : "v"(db), "v"(dc));
#define MF(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(y[(i) & 1]) : "v"(ha), "v"(hb));

template <int MODE>
__global__ __launch_bounds__(512) void bench(float* out, int iters)
{
    f2 a[16], b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    float s[32];
    double d[16], db = 0.999, dc = 0.001;
    f16x y[2];
    h8 ha, hb;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a[q] = f2{(float)threadIdx.x, (float)q}; d[q] = q; }
#pragma unroll
    for (int q = 0; q < 32; ++q) s[q] = q + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 16; ++q) { y[0][q] = 0.0f; y[1][q] = 0.0f; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { ha[q] = (_Float16)(0.001f * q); hb[q] = (_Float16)(0.002f * threadIdx.x); }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (MODE == 0 || MODE == 7) { PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) }
            if constexpr (MODE == 1 || MODE == 8) { SC(0) SC(1) SC(2) SC(3) SC(4) SC(5) SC(6) SC(7) SC(8) SC(9) SC(10) SC(11) SC(12) SC(13) SC(14) SC(15) }
            if constexpr (MODE == 2) { DP(0) DP(1) DP(2) DP(3) DP(4) DP(5) DP(6) DP(7) }
            if constexpr (MODE == 3) { MF(k) }
            if constexpr (MODE == 4) { MF(k) PK(8 * k) PK(8 * k + 1) PK(8 * k + 2) PK(8 * k + 3) PK(8 * k + 4) PK(8 * k + 5) PK(8 * k + 6) PK(8 * k + 7) }
            if constexpr (MODE == 5) { MF(k) SC(0) SC(1) SC(2) SC(3) SC(4) SC(5) SC(6) SC(7) SC(8) SC(9) SC(10) SC(11) SC(12) SC(13) SC(14) SC(15) }
            if constexpr (MODE == 6) { MF(k) DP(0) DP(1) DP(2) DP(3) DP(4) DP(5) DP(6) DP(7) }
        }
        if constexpr (MODE == 7 || MODE == 8) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { MF(k) }
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += a[q].x + a[q].y + (float)d[q] + y[0][q] + y[1][q];
#pragma unroll
    for (int q = 0; q < 32; ++q) acc += s[q];
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
int run(const char* what, float* d_out)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(512), 0, 0, d_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<MODE>, dim3(256), dim3(512), 0, 0, d_out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d  %-58s %8.1f ns per wave-iteration\n", MODE, what, ms * 1e6 / iters);
    return 0;
}

int main()
{
    float* d_out;
    CHECK(hipMalloc(&d_out, 4));
    run<0>("P: 128 v_pk_fma_f32", d_out);
    run<1>("S: 256 v_fma_f32 (same flops)", d_out);
    run<2>("D: 128 v_fma_f64", d_out);
    run<3>("M: 16 v_mfma_f32_32x32x16_f16", d_out);
    run<4>("P with M interleaved (1 MFMA per 8 packed)", d_out);
    run<5>("S with M interleaved (1 MFMA per 16 plain)", d_out);
    run<6>("D with M interleaved (1 MFMA per 8 f64)", d_out);
    run<7>("P then M", d_out);
    run<8>("S then M", d_out);
    return 0;
}
