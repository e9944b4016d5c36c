// VALU issue-rate microbenchmark for gfx950: how many cycles does a wave64 spend
// per instruction for the opcodes the FFT kernel is made of?  Decides whether
// packed-f32 code (v_pk_*) is worth the register shuffling it needs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int UNROLL = 8;      // independent chains
constexpr int INNER = 16;      // repeats of the UNROLL group per loop iteration

#define ASM8(op, fmt)                                                                    \
    asm volatile(op " " fmt "\n" : "+v"(a0) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a1) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a2) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a3) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a4) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a5) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a6) : "v"(b), "v"(c));                           \
    asm volatile(op " " fmt "\n" : "+v"(a7) : "v"(b), "v"(c));

template <int KIND>
__global__ __launch_bounds__(256) void bench(float* out, int iters)
{
    using f2 = __attribute__((ext_vector_type(2))) float;
    if constexpr (KIND == 0) {   // v_fma_f32
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0.999f, c = 0.001f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_fma_f32", "%0, %0, %1, %2") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if constexpr (KIND == 1) {   // v_pk_fma_f32
        f2 a0 = {1, 2}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, b = {0.999f, 0.999f}, c = {0.001f, 0.001f};
        a0.x = threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_pk_fma_f32", "%0, %0, %1, %2") }
        }
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else if constexpr (KIND == 2) {   // v_add_f32
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0.5f, c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_add_f32", "%0, %0, %1") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
    } else if constexpr (KIND == 3) {   // v_pk_add_f32
        f2 a0 = {1, 2}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, b = {0.5f, 0.5f}, c = {0, 0};
        a0.x = threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_pk_add_f32", "%0, %0, %1") }
        }
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else if constexpr (KIND == 4) {   // v_mov_b32 (chain through two regs)
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0.5f, c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_mov_b32", "%0, %1") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
    } else if constexpr (KIND == 5) {   // v_fma_f64
        double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0.999, c = 0.001;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_fma_f64", "%0, %0, %1, %2") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    } else if constexpr (KIND == 6) {   // v_cvt_f64_f32
        double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; float b = 0.999f, c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_cvt_f64_f32", "%0, %1") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + c;
    } else if constexpr (KIND == 7) {   // v_pk_mul_f32
        f2 a0 = {1, 2}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, b = {0.999f, 1.001f}, c = {0, 0};
        a0.x = threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_pk_mul_f32", "%0, %0, %1") }
        }
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else if constexpr (KIND == 8) {   // v_add_f64
        double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0.5, c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_add_f64", "%0, %0, %1") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c);
    } else if constexpr (KIND == 9) {   // v_cvt_f32_ubyte0
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; unsigned b = threadIdx.x * 2654435761u; float c = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < INNER; ++k) { ASM8("v_cvt_f32_ubyte0", "%0, %1") }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
    }
}

template <int KIND>
int run(const char* name, int waves_per_simd, float* d_out, double flops_per_inst)
{
    const int cus = 256;
    const int grid = cus * waves_per_simd;   // 256 threads = 4 waves = 1 wave per SIMD per block
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, d_out, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench<KIND>, dim3(grid), dim3(256), 0, 0, d_out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_wave = (double)iters * INNER * UNROLL;
    const double waves_per_simd_total = waves_per_simd;   // resident at once, all run concurrently
    // cycles per instruction per SIMD assuming 2.4 GHz
    const double ns_per_inst = ms * 1e6 / (insts_per_wave * waves_per_simd_total);
    const double tflops = flops_per_inst * 64 * insts_per_wave * 4.0 * grid / (ms * 1e-3) / 1e12;
    printf("%-18s waves/SIMD=%d  %.3f ms  %.3f ns/inst/SIMD (= %.2f clk @2.4GHz)  %.1f Tflop/s-equivalent\n",
           name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4, tflops);
    return 0;
}

int main()
{
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, d_out, 2);
        run<1>("v_pk_fma_f32", w, d_out, 4);
        run<2>("v_add_f32", w, d_out, 1);
        run<3>("v_pk_add_f32", w, d_out, 2);
        run<7>("v_pk_mul_f32", w, d_out, 2);
        run<4>("v_mov_b32", w, d_out, 0);
        run<5>("v_fma_f64", w, d_out, 2);
        run<8>("v_add_f64", w, d_out, 1);
        run<6>("v_cvt_f64_f32", w, d_out, 0);
        run<9>("v_cvt_f32_ubyte0", w, d_out, 0);
    }
    return 0;
}
