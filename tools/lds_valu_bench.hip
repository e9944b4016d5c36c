// Do a wave's LDS exchanges hide under another wave's packed-f32 arithmetic on gfx950?
// Mimics one frame of the fused N=4096 kernel per loop iteration and wave:
//   V = 320 v_pk_fma_f32 (16 independent chains), W = 32 ds_write_b64, R = 48 ds_read_b64
// (conflict-free, `s_waitcnt lgkmcnt(0)` after every group of 16 like the exchange phases).
//   mode 0: V only            mode 1: W+R only          mode 2: V, W, R in sequence per wave
//   mode 3: as 2 with a workgroup barrier after the stores and after the loads
//   mode 4: specialised -- the first half of the workgroup's waves do 2 x V, the second half
//           2 x (W+R): one arithmetic wave beside one LDS wave on every SIMD
//   mode 5: V and W/R interleaved instruction by instruction in one wave (4 pk : 1 ds)
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_valu_bench.hip -o tools/lds_valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define PK16(a, b, c)                                                   \
    _Pragma("unroll") for (int q = 0; q < 16; ++q)                      \
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[q]) : "v"(b), "v"(c));

template <int MODE, int WG>
__global__ __launch_bounds__(WG) void bench(float* out, int iters, unsigned long long* cycles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per-wave private region of 16 x 64 x 8 B + padding: conflict-free b64 accesses
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    using lds_f2 = volatile __attribute__((address_space(3))) f2;
    lds_f2* const mine = (lds_f2*)(smem) + wave * (16 * 65) + lane;
    f2 a[16], b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = f2{(float)threadIdx.x, (float)q};
    f2 r[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = a[q];
    const bool valu_wave = wave < (WG / 128);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll 1
            for (int k = 0; k < 20; ++k) { PK16(a, b, c) }
        }
        if constexpr (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll 1
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int q = 0; q < 16; ++q) mine[q * 65] = a[q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (MODE == 3) asm volatile("s_barrier" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) r[q] = mine[q * 65];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (MODE == 3) asm volatile("s_barrier" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] += r[q] * 1e-30f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) r[q] = mine[q * 65];       // the third fetch (twiddles / raw)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] += r[q] * 1e-30f;
        }
        if constexpr (MODE == 4) {
            if (valu_wave) {
#pragma unroll 1
                for (int k = 0; k < 40; ++k) { PK16(a, b, c) }
            } else {
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) mine[q * 65] = a[q];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 16; ++q) r[q] = mine[q * 65];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (g & 1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) r[q] = mine[q * 65];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[q] += r[q] * 1e-30f;
                }
            }
        }
        if constexpr (MODE == 5) {
            // 80 DS operations spread between the 320 packed FMAs (4 : 1), consumed a group later
#pragma unroll
            for (int g = 0; g < 5; ++g) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (g == 0 || g == 2) mine[q * 65] = a[q];
                    else r[q] = mine[q * 65];
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(q + 1) & 15]) : "v"(b), "v"(c));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(q + 5) & 15]) : "v"(b), "v"(c));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(q + 9) & 15]) : "v"(b), "v"(c));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(q + 13) & 15]) : "v"(b), "v"(c));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (g == 1 || g == 3 || g == 4) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[q] += r[q] * 1e-30f;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f2 s = {0, 0};
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[q];
    out[blockIdx.x * WG + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE, int WG>
int run(const char* what, int wgs_per_cu, float* d_out, unsigned long long* d_cyc)
{
    const int grid = 256 * wgs_per_cu, iters = 400;
    const int lds = (WG / 64) * 16 * 65 * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench<MODE, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench<MODE, WG>), dim3(grid), dim3(WG), lds, 0, d_out, 20, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((bench<MODE, WG>), dim3(grid), dim3(WG), lds, 0, d_out, iters, d_cyc);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc = 0;
    CHECK(hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost));
    const int waves_per_simd = wgs_per_cu * WG / 256;
    printf("mode %d %-34s WG=%4d x%d/CU (%d waves/SIMD): %8.1f ns per iteration = %6.0f cycles (s_memtime), per wave-iteration per SIMD %6.0f cycles\n",
           MODE, what, WG, wgs_per_cu, waves_per_simd, ms * 1e6 / iters, (double)cyc / iters, (double)cyc / iters / waves_per_simd);
    return 0;
}

int main()
{
    float* d_out;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 4 * 1024));
    CHECK(hipMalloc(&d_cyc, 8));
    for (int w = 1; w <= 4; ++w) {
        run<0, 256>("VALU only (320 pk_fma)", w, d_out, d_cyc);
        run<1, 256>("LDS only (32 st + 48 ld b64)", w, d_out, d_cyc);
        run<2, 256>("VALU then LDS, per wave", w, d_out, d_cyc);
        run<5, 256>("VALU and LDS interleaved", w, d_out, d_cyc);
    }
    run<0, 512>("VALU only", 1, d_out, d_cyc);
    run<1, 512>("LDS only", 1, d_out, d_cyc);
    run<2, 512>("VALU then LDS", 1, d_out, d_cyc);
    run<3, 512>("VALU then LDS + barriers", 1, d_out, d_cyc);
    run<4, 512>("specialised 2xV | 2xLDS (2 iter-equivalents)", 1, d_out, d_cyc);
    run<5, 512>("interleaved", 1, d_out, d_cyc);
    run<3, 256>("VALU then LDS + barriers", 3, d_out, d_cyc);
    run<4, 1024>("specialised, 4 waves/SIMD", 1, d_out, d_cyc);
    return 0;
}
