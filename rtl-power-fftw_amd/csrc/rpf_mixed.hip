// rpf_mixed.hip -- KM: LDS-resident mixed-radix kernels for the "round" sizes people actually
// type (500 -- the man page's own example, doc/rtl_power_fftw.1.md:182 --, 1000, 1200, 2000, 3000,
// 5000, 7000, 10000, 15000, 20000, 50000 ...) and for 16384 and 32768.  Bluestein serves such sizes
// with two power-of-two transforms of 2-4 N points each; a transform of the length itself costs a
// fifth of that.
//
// Three kernels:
//  * mixed_plan_kernel (mixed_plan_kernels.h, mixed_core.h, dft_small.h): K1's scheme -- in place by element name
//    through one padded LDS slab, radices 2 ... 25 (two to four passes), twiddles and the f64
//    accumulators in registers for the whole launch -- compiled for the 145 sizes of
//    mixed_plans.inc (N <= 16384) with the plan (radices, butterflies per thread, frame slots per
//    workgroup, twiddle placement) that measured fastest on the GPU (tools/gen_mixed_plans.py,
//    tools/pick_mixed_plans.py, profiles/r02_mixed_plan_search.txt).  0.25 ... 1.06 Tsample/s.
//  * mixed_split_kernel (same header; its tables are compiled in rpf_mixed_split.hip): N = P M, P = 2 ... 5, M even
//    with factors 2 ... 25 (mixed_plans_split.inc: 54 sizes, 10500 ... 108000): workgroup b computes the residue p of the spectrum,
//    X[p + P k] = FFT_M(x'_p)[k], so only raw bytes cross workgroups.  0.14 ... 0.47 Tsample/s.
//  * mixed_kernel: any other even N <= 5120 with prime factors 2, 3, 5, runtime plan: Stockham
//    autosort, decimation in frequency, one radix (5, 4, 3, 2) per pass, natural order in and
//    out, two LDS buffers per frame slot:
//
//   pass with sub-length n = N / s, n1 = n / r, butterfly (p < n1, q < s):
//       y[q + s (r p + j)] = W_n^{p j} * sum_k x[q + s (p + k n1)] W_r^{j k},     W_n^{p j} = W_N^{p j s}
//
//    The first pass reads the u8 samples straight from HBM (2-byte loads, coalesced across the
//    threads of a frame) and applies (v - 127) (-1)^n [window] exactly like K1; the last pass
//    leaves |X|^2 in double accumulators that live in LDS for the whole launch (one owner thread
//    per bin: plain read-modify-write).  TPF threads per frame (a power of two, about N/4),
//    WG / TPF frames side by side in a 256-thread workgroup.  0.1 ... 0.3 Tsample/s.
// HBM traffic of all three = the input bytes (2N per frame; P times that from L2 in the split
// form); bound by VALU + LDS like K1.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mixed_plan_kernels.h"

namespace rpf {

namespace {

constexpr int kMixedWG = 256;
constexpr int kMaxFactors = 12;

struct MixedPlan {
    int nfac;
    int radix[kMaxFactors];
};

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
template <int R, bool FIRST, bool LAST>
__device__ __forceinline__ void mixed_pass(const cf* __restrict__ x, cf* __restrict__ y, const uint8_t* __restrict__ frame,
                                           const float* __restrict__ window, const cf* __restrict__ tw, double* acc,
                                           bool active, int N, int s, int n1, int t, int tpf)
{
    const float inv_s = 1.0f / static_cast<float>(s);
    for (int b = t; b < N / R; b += tpf) {
        const int p = FIRST ? b : div_small(b, s, inv_s), q = FIRST ? 0 : b - p * s;
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
        SmallDft<R>::run(v);
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

// variant 0 of a size is what ships (mixed_plans.inc, picked from GPU timings of the candidates by
// tools/pick_mixed_plans.py); the candidates themselves (tools/gen_mixed_plans.py) exist in the
// tuning build only (rpf_mixed_split.hip holds them, after the split form's table).
const PlanEntry kPlans[] = {
#include "mixed_plans.inc"
};
const PlanEntry* find_plan(int N, int variant)
{
    for (const PlanEntry& e : kPlans)
        if (e.N == N && e.variant == variant) return &e;
    int n = 0;
    const PlanEntry* split = split_plan_table(&n);
    for (int i = 0; i < n; ++i)
        if (split[i].N == N && split[i].variant == variant) return &split[i];
    return nullptr;
}
// how size N runs (nullptr: not a planned size; fn == nullptr: a tuning-build candidate without a windowed twin)
const PlanForm* find_form(int N, int variant, bool windowed)
{
    if (variant == 0) {
        int n = 0;
        const FormOverride* o = form_override_table(&n);
        for (int i = 0; i < n; ++i)
            if (o[i].N == N && o[i].windowed == windowed) return &o[i].form;
    }
    const PlanEntry* pe = find_plan(N, variant);
    return pe ? &pe->form(windowed) : nullptr;
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

int lds_bytes(int N)       // [frame slots][2 buffers + accumulators] + the twiddle table
{
    return (kMixedWG / threads_per_frame(N)) * N * (2 * (int)sizeof(cf) + (int)sizeof(double)) + N * (int)sizeof(cf);
}

}  // namespace

// Even N served from one workgroup's LDS.  The single source of truth for WHICH sizes is the pair of
// tables the library is compiled from: mixed_plans.inc (the planned kernel: round sizes with small prime
// factors, and 16384 -- the one power of two K1 cannot hold and the four-step kernels serve at half the
// rate) and mixed_plans_split.inc (the split form N = P x M); mixed_plans_override.inc moves single runs
// (plain or windowed) of those sizes onto another form.  The run-time Stockham kernel takes every other
// even size up to 5120 bins with prime factors 2, 3, 5 only.
// variant != 0 (tuning build): another plan of the same size; 100 = the Stockham kernel for a size that
// has a plan.
#ifdef RPF_TUNING
constexpr int kStockhamVariant = 100;
#endif
bool mixed_supported(int N, int variant)
{
    if (N < 2 || (N & 1)) return false;
    if (find_plan(N, variant)) return true;       // (16384 is one of these; the tuning build can time K1's sizes, too)
    if ((N & (N - 1)) == 0) return false;
#ifndef RPF_TUNING
    if (variant != 0) return false;
#else
    if (variant != 0 && variant != kStockhamVariant) return false;
#endif
    MixedPlan plan;
    return factorise(N, &plan) && lds_bytes(N) <= 160 * 1024;
}

hipError_t plan_mixed(int N, int variant, bool windowed, int device, LaunchInfo* li)
{
    if (!mixed_supported(N, variant)) return hipErrorInvalidValue;
    const PlanForm* pf = find_form(N, variant, windowed);
    if (pf && !pf->fn) return hipErrorInvalidValue;      // (a tuning-build candidate without its windowed twin)
    const int lds = pf ? pf->lds : lds_bytes(N);
    const int wg = pf ? pf->wg : kMixedWG;
    const void* fn = pf ? reinterpret_cast<const void*>(pf->fn) : reinterpret_cast<const void*>(mixed_kernel);
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, wg, lds);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    li->grid = std::max(per_cu, 1) * prop.multiProcessorCount;
    if (pf && pf->split > 1) {
        // split form: whole rounds of the 8 XCDs (its XCD-local mapping) where the device has that many
        // resident workgroups, else any multiple of the split factor; a device too small for even one
        // group gets no plan, and rpf_engine_create takes the next kernel family
        const int unit = li->grid >= 8 * pf->split ? 8 * pf->split : pf->split;
        li->grid -= li->grid % unit;
        if (li->grid < pf->split) return hipErrorInvalidValue;
    }
    li->block = wg;
    li->fpw = pf ? pf->fpw : kMixedWG / threads_per_frame(N);
    li->lds_bytes = lds;
    return hipSuccess;
}

// d_twN: master twiddles W_N^k (make_twiddles); one partial spectrum of N doubles per workgroup.
hipError_t launch_mixed(int N, int variant, const uint8_t* d_stream, long nframes, const cf* d_twN, const float* d_window,
                        double* d_partial, int max_grid, hipStream_t stream, LaunchInfo* li)
{
    if (!mixed_supported(N, variant) || max_grid < 1 || nframes < 1) return hipErrorInvalidValue;
    int wg = kMixedWG, fpw = 0, lds = 0, grid = 0, split = 1;
    if (const PlanForm* pform = find_form(N, variant, d_window != nullptr)) {
        const PlanForm& pf = *pform;
        if (!pf.fn) return hipErrorInvalidValue;      // (a tuning-build candidate without its windowed twin)
        wg = pf.wg, fpw = pf.fpw, lds = pf.lds, split = pf.split;
        // no more workgroups than frames to share out (x the split factor: one workgroup per residue)
        long groups = std::min<long>(max_grid / split, (nframes + fpw - 1) / fpw);
        if (split > 1 && groups > 8) groups -= groups % 8;        // whole rounds of the 8 XCDs
        if (groups < 1) return hipErrorInvalidValue;
        grid = static_cast<int>(groups) * split;
        hipLaunchKernelGGL(pf.fn, dim3(grid), dim3(wg), lds, stream, d_stream, nframes, d_twN, d_window, d_partial);
    } else {
        MixedPlan plan;
        if (!factorise(N, &plan)) return hipErrorInvalidValue;
        const int tpf = threads_per_frame(N);
        fpw = kMixedWG / tpf, lds = lds_bytes(N);
        grid = static_cast<int>(std::min<long>(max_grid, (nframes + fpw - 1) / fpw));
        hipLaunchKernelGGL(mixed_kernel, dim3(grid), dim3(kMixedWG), lds, stream, d_stream, nframes, N, tpf, plan, d_twN,
                           d_window, d_partial);
    }
    if (li) {
        li->grid = grid;
        li->block = wg;
        li->fpw = fpw;
        li->lds_bytes = lds;
        li->slots = grid / split;
    }
    return hipGetLastError();
}

}  // namespace rpf
