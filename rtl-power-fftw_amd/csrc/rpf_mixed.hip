// rpf_mixed.hip -- KM: LDS-resident mixed-radix kernel for the "round" sizes people actually
// type: even N <= 4096 whose prime factors are 2, 3 and 5 and that are not powers of two
// (500 -- the man page's own example, doc/rtl_power_fftw.1.md:182 --, 1000, 1200, 1500, 2000,
// 3000, 3600, 4000, ...).  Bluestein (KB) serves such sizes with two power-of-two transforms of
// 2-4 N points each; a transform of the length itself costs a fifth of that.
//
// Stockham autosort, decimation in frequency, one radix per pass (radices 5, 4, 3, 2 in the
// order the host picks), natural order in and out, two LDS buffers per frame slot:
//
//   pass with sub-length n = N / s, n1 = n / r, butterfly (p < n1, q < s):
//       y[q + s (r p + j)] = W_n^{p j} * sum_k x[q + s (p + k n1)] W_r^{j k},     W_n^{p j} = W_N^{p j s}
//
// The first pass reads the u8 samples straight from HBM (2-byte loads, coalesced across the
// threads of a frame) and applies (v - 127) (-1)^n [window] exactly like K1; the last pass leaves
// the spectrum in natural order in LDS, and |X|^2 goes into double accumulators that live in
// LDS for the whole launch (one owner thread per bin: plain read-modify-write).  TPF threads per
// frame (a power of two, about N/4), WG / TPF frames side by side in a 256-thread workgroup.
// HBM traffic = the 2N input bytes per frame; bound by VALU + LDS like K1, less tuned than K1.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

constexpr int kMixedWG = 256;
constexpr int kMaxFactors = 12;

struct MixedPlan {
    int nfac;
    int radix[kMaxFactors];
};

// r-point DFT of v[0..r) in place (forward, e^{-2 pi i / r}); constants correctly rounded floats.
template <int R>
__device__ __forceinline__ void small_dft(cf* v);

template <>
__device__ __forceinline__ void small_dft<2>(cf* v)
{
    const cf a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
}

template <>
__device__ __forceinline__ void small_dft<4>(cf* v)
{
    Dft<4>::run(v);
}

template <>
__device__ __forceinline__ void small_dft<3>(cf* v)
{
    constexpr float kS3 = 0.86602540378443864676f;       // sin(2 pi / 3)
    const cf s = v[1] + v[2], d = v[1] - v[2];
    const cf m = v[0] - s * 0.5f;                         // v0 + cos(2 pi/3) (v1 + v2)
    const cf jd = mul_mi(d) * kS3;                        // -i sin(2 pi/3) (v1 - v2)
    v[0] = v[0] + s;
    v[1] = m + jd;
    v[2] = m - jd;
}

template <>
__device__ __forceinline__ void small_dft<5>(cf* v)
{
    constexpr float kC1 = 0.30901699437494742410f;       // cos(2 pi / 5)
    constexpr float kC2 = -0.80901699437494742410f;      // cos(4 pi / 5)
    constexpr float kS1 = 0.95105651629515357212f;       // sin(2 pi / 5)
    constexpr float kS2 = 0.58778525229247312917f;       // sin(4 pi / 5)
    const cf s14 = v[1] + v[4], d14 = v[1] - v[4];
    const cf s23 = v[2] + v[3], d23 = v[2] - v[3];
    const cf a1 = v[0] + s14 * kC1 + s23 * kC2;
    const cf a2 = v[0] + s14 * kC2 + s23 * kC1;
    const cf b1 = mul_mi(d14 * kS1 + d23 * kS2);          // -i (...)
    const cf b2 = mul_mi(d14 * kS2 - d23 * kS1);
    v[0] = v[0] + s14 + s23;
    v[1] = a1 + b1;
    v[4] = a1 - b1;
    v[2] = a2 + b2;
    v[3] = a2 - b2;
}

// b / s for 0 <= b < 4096, 1 <= s <= 4096 without an integer division (s is not a power of two)
__device__ __forceinline__ int div_small(int b, int s, float inv_s)
{
    int p = static_cast<int>(static_cast<float>(b) * inv_s);
    p -= (p * s > b);
    p += ((p + 1) * s <= b);
    return p;
}

// One pass of radix R over one frame: x (LDS, or the raw stream when FIRST) -> y (LDS), or, in
// the LAST pass, straight into the double accumulators (a butterfly owns the same R bins in every
// frame, so the spectrum never goes back to LDS).  tw: W_N^k in LDS.
// CN/CS/CN1/CTPF > 0: N, s, n1 and the threads per frame are compile-time constants (the
// specialised kernels below): strides fold into DS immediates, b / s into a multiply-shift,
// and the butterfly loop unrolls -- the generic form spends 3/4 of its instructions on indices.
template <int R, bool FIRST, bool LAST, int CN = 0, int CS = 0, int CN1 = 0, int CTPF = 0>
__device__ __forceinline__ void mixed_pass(const cf* __restrict__ x, cf* __restrict__ y, const uint8_t* __restrict__ frame,
                                           const float* __restrict__ window, const cf* __restrict__ tw, double* acc,
                                           bool active, int N_, int s_, int n1_, int t, int tpf_)
{
    const int N = CN ? CN : N_, s = CS ? CS : s_, n1 = CN1 ? CN1 : n1_, tpf = CTPF ? CTPF : tpf_;
    const float inv_s = 1.0f / static_cast<float>(s);
    auto butterfly = [&](int b) {
        const int p = FIRST ? b : (CS ? b / (CS ? CS : 1) : div_small(b, s, inv_s)), q = FIRST ? 0 : b - p * s;
        cf v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if constexpr (FIRST) {
                const int n = p + k * n1;                                           // s = 1
                const uint32_t iq = *reinterpret_cast<const uint16_t*>(frame + 2 * n);
                const cf f = iq_plus_2p23(iq) - (kTwo23 + 127.0f);                  // exact (datastore.cxx:75)
                const float sg = (n & 1) ? -1.0f : 1.0f;                            // datastore.cxx:73,76-77
                v[k] = window ? f * (window[n] * sg) : f * sg;
            } else {
                v[k] = x[q + s * (p + k * n1)];
            }
        }
        small_dft<R>(v);
        const int o = q + s * (R * p);
        if constexpr (LAST) {
            // n1 = 1, p = 0: no twiddles; pwr += Re^2 + Im^2 in double (datastore.cxx:83-85)
            if (active) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double re = static_cast<double>(v[j].x), im = static_cast<double>(v[j].y);
                    acc[o + s * j] = __builtin_fma(im, im, __builtin_fma(re, re, acc[o + s * j]));
                }
            }
        } else {
            y[o] = v[0];
            if (p == 0) {
#pragma unroll
                for (int j = 1; j < R; ++j) y[o + s * j] = v[j];
            } else {
#pragma unroll
                for (int j = 1; j < R; ++j) y[o + s * j] = cmul(v[j], tw[p * j * s]);   // W_n^{p j} = W_N^{p j s}, p j s < N
            }
        }
    };
    if constexpr (CN > 0) {
        constexpr int BFLY = CN / R, ITER = (BFLY + CTPF - 1) / CTPF;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int b = t + i * CTPF;
            if ((i + 1) * CTPF <= BFLY || b < BFLY) butterfly(b);
        }
    } else {
        for (int b = t; b < N / R; b += tpf) butterfly(b);
    }
}

template <bool FIRST, bool LAST>
__device__ __forceinline__ void mixed_pass_r(int r, const cf* x, cf* y, const uint8_t* frame, const float* window,
                                             const cf* tw, double* acc, bool active, int N, int s, int n1, int t, int tpf)
{
    switch (r) {
        case 5: mixed_pass<5, FIRST, LAST>(x, y, frame, window, tw, acc, active, N, s, n1, t, tpf); break;
        case 4: mixed_pass<4, FIRST, LAST>(x, y, frame, window, tw, acc, active, N, s, n1, t, tpf); break;
        case 3: mixed_pass<3, FIRST, LAST>(x, y, frame, window, tw, acc, active, N, s, n1, t, tpf); break;
        default: mixed_pass<2, FIRST, LAST>(x, y, frame, window, tw, acc, active, N, s, n1, t, tpf); break;
    }
}

// (Staging the next frame's bytes in LDS by LDS-DMA while the current one is transformed was
// measured 20-25 % SLOWER: with 4-6 workgroups per CU the 2-byte loads' latency is already hidden,
// and the extra barrier and LDS footprint cost more.)
__global__ __launch_bounds__(kMixedWG) void mixed_kernel(const uint8_t* __restrict__ stream, long nframes, int N, int tpf,
                                                        MixedPlan plan, const cf* __restrict__ twN,
                                                        const float* __restrict__ window, double* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int fpw = kMixedWG / tpf;
    const int fs = tid / tpf, t = tid % tpf;
    cf* const bufA = reinterpret_cast<cf*>(smem) + static_cast<size_t>(fs) * 2 * N;
    cf* const bufB = bufA + N;
    double* const acc_all = reinterpret_cast<double*>(smem + static_cast<size_t>(fpw) * 2 * N * sizeof(cf));
    double* const acc = acc_all + static_cast<size_t>(fs) * N;
    cf* const tw = reinterpret_cast<cf*>(acc_all + static_cast<size_t>(fpw) * N);      // W_N^k, k < N
    for (int bin = t; bin < N; bin += tpf) acc[bin] = 0.0;
    for (int k = tid; k < N; k += kMixedWG) tw[k] = twN[k];
    __syncthreads();

    const long stride = static_cast<long>(gridDim.x) * fpw;
    for (long fb = static_cast<long>(blockIdx.x) * fpw; fb < nframes; fb += stride) {
        const bool active = (fb + fs) < nframes;
        const uint8_t* const frame = stream + (active ? fb + fs : nframes - 1) * 2 * static_cast<long>(N);
        int s = 1, n = N;
        cf* src = bufB;
        cf* dst = bufA;
        for (int pass = 0; pass < plan.nfac; ++pass) {
            const int r = plan.radix[pass], n1 = n / r;
            const bool last = pass == plan.nfac - 1;
            if (pass == 0) {
                if (last) mixed_pass_r<true, true>(r, src, dst, frame, window, tw, acc, active, N, s, n1, t, tpf);
                else mixed_pass_r<true, false>(r, src, dst, frame, window, tw, acc, active, N, s, n1, t, tpf);
            } else {
                if (last) mixed_pass_r<false, true>(r, src, dst, frame, window, tw, acc, active, N, s, n1, t, tpf);
                else mixed_pass_r<false, false>(r, src, dst, frame, window, tw, acc, active, N, s, n1, t, tpf);
            }
            s *= r;
            n = n1;
            cf* const tmp = src;
            src = dst;
            dst = tmp;
            if (!last) __syncthreads();
        }
        // (the last pass reads the buffer the next frame's first pass does NOT write when the number of
        // passes is even, and the one it does write when it is odd: one barrier covers both)
        __syncthreads();
    }
    __syncthreads();
    for (int bin = tid; bin < N; bin += kMixedWG) {
        double v = 0.0;
        for (int k = 0; k < fpw; ++k) v += acc_all[static_cast<size_t>(k) * N + bin];
        partial[static_cast<size_t>(blockIdx.x) * N + bin] = v;
    }
}

// ---- specialised kernels: every constant known at compile time --------------------------------
constexpr int threads_per_frame_c(int N)
{
    int tpf = 64;
    while (tpf < 256 && tpf < N / 4) tpf *= 2;
    return tpf;
}

// the twiddle table sits in LDS unless leaving it in HBM/L1 lets one more workgroup onto the CU
// (measured: N = 3000 158 -> 240 Gsample/s without it, N = 4000 171 -> 114)
constexpr bool spec_tw_in_lds(int N)
{
    const int slots = kMixedWG / threads_per_frame_c(N);
    const int base = slots * N * (2 * (int)sizeof(cf) + (int)sizeof(double));
    return (160 * 1024) / (base + N * (int)sizeof(cf)) >= (160 * 1024) / base;
}

template <int N, int TPF, int S, bool FIRST, int R, int... Rest>
__device__ __forceinline__ void spec_passes(cf* src, cf* dst, const uint8_t* frame, const float* window, const cf* tw,
                                            double* acc, bool active, int t)
{
    constexpr int n1 = N / S / R;
    constexpr bool last = sizeof...(Rest) == 0;
    mixed_pass<R, FIRST, last, N, S, n1, TPF>(src, dst, frame, window, tw, acc, active, N, S, n1, t, TPF);
    if constexpr (!last) {
        __syncthreads();
        spec_passes<N, TPF, S * R, false, Rest...>(dst, src, frame, window, tw, acc, active, t);
    }
}

template <int N, int... Rs>
__global__ __launch_bounds__(kMixedWG) void mixed_spec_kernel(const uint8_t* __restrict__ stream, long nframes,
                                                             const cf* __restrict__ twN, const float* __restrict__ window,
                                                             double* __restrict__ partial)
{
    constexpr int TPF = threads_per_frame_c(N), FPW = kMixedWG / TPF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int fs = tid / TPF, t = tid % TPF;
    cf* const bufA = reinterpret_cast<cf*>(smem) + fs * 2 * N;
    cf* const bufB = bufA + N;
    double* const acc_all = reinterpret_cast<double*>(smem + FPW * 2 * N * sizeof(cf));
    double* const acc = acc_all + fs * N;
    constexpr bool TWLDS = spec_tw_in_lds(N);
    cf* const twl = reinterpret_cast<cf*>(acc_all + FPW * N);
    const cf* const tw = TWLDS ? twl : twN;
    for (int bin = t; bin < N; bin += TPF) acc[bin] = 0.0;
    if constexpr (TWLDS)
        for (int k = tid; k < N; k += kMixedWG) twl[k] = twN[k];
    __syncthreads();
    const long stride = static_cast<long>(gridDim.x) * FPW;
#pragma unroll 1
    for (long fb = static_cast<long>(blockIdx.x) * FPW; fb < nframes; fb += stride) {
        const bool active = (fb + fs) < nframes;
        const uint8_t* const frame = stream + (active ? fb + fs : nframes - 1) * (2L * N);
        spec_passes<N, TPF, 1, true, Rs...>(bufB, bufA, frame, window, tw, acc, active, t);
        __syncthreads();
    }
    __syncthreads();
    for (int bin = tid; bin < N; bin += kMixedWG) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < FPW; ++k) v += acc_all[k * N + bin];
        partial[static_cast<size_t>(blockIdx.x) * N + bin] = v;
    }
}

using SpecFn = void (*)(const uint8_t*, long, const cf*, const float*, double*);
struct SpecEntry {
    int N;
    SpecFn fn;
};
// The sizes people type; any other 5-smooth size runs on the generic kernel above.
const SpecEntry kSpec[] = {
    {100, mixed_spec_kernel<100, 5, 5, 4>},       {200, mixed_spec_kernel<200, 5, 5, 4, 2>},
    {250, mixed_spec_kernel<250, 5, 5, 5, 2>},    {300, mixed_spec_kernel<300, 5, 5, 4, 3>},
    {400, mixed_spec_kernel<400, 5, 5, 4, 4>},    {500, mixed_spec_kernel<500, 5, 5, 5, 4>},
    {600, mixed_spec_kernel<600, 5, 5, 4, 3, 2>}, {800, mixed_spec_kernel<800, 5, 5, 4, 4, 2>},
    {1000, mixed_spec_kernel<1000, 5, 5, 5, 4, 2>},    {1200, mixed_spec_kernel<1200, 5, 5, 4, 4, 3>},
    {1500, mixed_spec_kernel<1500, 5, 5, 5, 4, 3>},    {1600, mixed_spec_kernel<1600, 5, 5, 4, 4, 4>},
    {2000, mixed_spec_kernel<2000, 5, 5, 5, 4, 4>},    {2400, mixed_spec_kernel<2400, 5, 5, 4, 4, 3, 2>},
    {2500, mixed_spec_kernel<2500, 5, 5, 5, 5, 4>},    {3000, mixed_spec_kernel<3000, 5, 5, 5, 4, 3, 2>},
    {3200, mixed_spec_kernel<3200, 5, 5, 4, 4, 4, 2>}, {3600, mixed_spec_kernel<3600, 5, 5, 4, 4, 3, 3>},
    {4000, mixed_spec_kernel<4000, 5, 5, 5, 4, 4, 2>}, {4800, mixed_spec_kernel<4800, 5, 5, 4, 4, 4, 3>},
    {5000, mixed_spec_kernel<5000, 5, 5, 5, 5, 4, 2>}, {6000, mixed_spec_kernel<6000, 5, 5, 5, 4, 4, 3>},
    {6400, mixed_spec_kernel<6400, 5, 5, 4, 4, 4, 4>},
};
const SpecEntry* find_spec(int N)
{
    for (const SpecEntry& e : kSpec)
        if (e.N == N) return &e;
    return nullptr;
}

bool factorise(int N, MixedPlan* plan)
{
    int rest = N, nf = 0;
    const int radices[] = {5, 4, 3, 2};
    for (int r : radices)
        while (rest % r == 0 && rest > 1) {
            if (nf >= kMaxFactors) return false;
            plan->radix[nf++] = r;
            rest /= r;
        }
    plan->nfac = nf;
    return rest == 1 && nf > 0;
}

int threads_per_frame(int N)
{
    int tpf = 64;
    while (tpf < 256 && tpf < N / 4) tpf *= 2;
    return tpf;
}

int lds_bytes(int N)       // [frame slots][2 buffers + accumulators] (+ the twiddle table)
{
    const bool table = !find_spec(N) || spec_tw_in_lds(N);
    return (kMixedWG / threads_per_frame(N)) * N * (2 * (int)sizeof(cf) + (int)sizeof(double)) +
           (table ? N * (int)sizeof(cf) : 0);
}

}  // namespace

// even N whose frame (two buffers + accumulators [+ twiddles]) fits one workgroup's LDS -- up to 5120
// bins for any such size, 6400 for the specialised ones --, only prime factors 2, 3, 5, not a power
// of two (those are K1's)
bool mixed_supported(int N)
{
    if (N < 2 || (N & 1) || N > 6400 || (N & (N - 1)) == 0) return false;
    MixedPlan plan;
    return factorise(N, &plan) && lds_bytes(N) <= 160 * 1024;
}

hipError_t plan_mixed(int N, int device, LaunchInfo* li)
{
    if (!mixed_supported(N)) return hipErrorInvalidValue;
    const int lds = lds_bytes(N);
    const SpecEntry* spec = find_spec(N);
    const void* fn = spec ? reinterpret_cast<const void*>(spec->fn) : reinterpret_cast<const void*>(mixed_kernel);
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kMixedWG, lds);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    li->grid = std::max(per_cu, 1) * prop.multiProcessorCount;
    li->block = kMixedWG;
    li->fpw = kMixedWG / threads_per_frame(N);
    li->lds_bytes = lds;
    return hipSuccess;
}

// d_twN: master twiddles W_N^k (make_twiddles); one partial spectrum of N doubles per workgroup.
hipError_t launch_mixed(int N, const uint8_t* d_stream, long nframes, const cf* d_twN, const float* d_window,
                        double* d_partial, int grid, hipStream_t stream, LaunchInfo* li)
{
    MixedPlan plan;
    if (!mixed_supported(N) || !factorise(N, &plan) || grid < 1) return hipErrorInvalidValue;
    const int tpf = threads_per_frame(N), lds = lds_bytes(N);
    if (const SpecEntry* spec = find_spec(N))
        hipLaunchKernelGGL(spec->fn, dim3(grid), dim3(kMixedWG), lds, stream, d_stream, nframes, d_twN, d_window, d_partial);
    else
        hipLaunchKernelGGL(mixed_kernel, dim3(grid), dim3(kMixedWG), lds, stream, d_stream, nframes, N, tpf, plan, d_twN,
                           d_window, d_partial);
    if (li) {
        li->grid = grid;
        li->block = kMixedWG;
        li->fpw = kMixedWG / tpf;
        li->lds_bytes = lds;
    }
    return hipGetLastError();
}

}  // namespace rpf
