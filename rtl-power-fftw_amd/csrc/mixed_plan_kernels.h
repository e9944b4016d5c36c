// mixed_plan_kernels.h -- the planned mixed-radix kernel, its split form, and the table entries that name their
// instantiations.  Included by the two translation units the tables are compiled in (rpf_mixed.hip: the planned sizes
// of mixed_plans.inc; rpf_mixed_split.hip: the split form's sizes and the per-size overrides), so that the ~ 500
// kernels build in parallel; the kernels themselves are internal to each unit.
#pragma once

#include <hip/hip_runtime.h>

#include "mixed_core.h"
#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

using PlanFn = void (*)(const uint8_t*, long, const cf*, const float*, double*);
// One way of running a size: the kernel, its workgroup, frames per workgroup, LDS, and the split factor (1: the
// planned kernel; P: the split form, N = P x the plan's length).  A size has one for plain runs and one for windowed
// runs -- not necessarily the same plan: the window wants registers or LDS the fastest plain plan may not have left.
struct PlanForm {
    PlanFn fn;
    int wg, fpw, lds, split;
};
struct PlanEntry {
    int N, variant;
    PlanForm plain, windowed;
    const PlanForm& form(bool w) const { return w ? windowed : plain; }
};
// Runs of the shipped sizes (variant 0) that do better on another form than their table entry gives them --
// mostly windowed runs, whose window values want registers or LDS the fastest plain plan has not left:
// mixed_plans_override.inc, picked from GPU timings (tools/gen_mixed_plans.py winsearch / splitsearch).
struct FormOverride {
    int N;
    bool windowed;
    PlanForm form;
};
// rpf_mixed_split.hip: the split form's table (mixed_plans_split.inc; the tuning build's candidates after it);
// rpf_mixed_override.hip: the overrides
const PlanEntry* split_plan_table(int* count);
const FormOverride* form_override_table(int* count);

namespace {

// ---- planned kernels (mixed_core.h): K1's scheme for composite lengths -------------------------
// In place by element name through one padded slab per frame slot, composite radices up to 25
// (two or three passes where the Stockham kernels above take four to seven), twiddles and
// accumulators in registers for the whole launch, the next frame's samples prefetched into
// registers while the current one is transformed.
// TS: stride of W_{PL::N}^k in the table (1; P in the split form, whose table is W_{P N})
template <class PL, int I, int TS = 1>
__device__ __forceinline__ void plan_load_twiddles(int t, const cf* __restrict__ twN, cf* tw)
{
    if constexpr (I < PL::F - 1 && (PL::TW == 0 || I == 0)) {
        if (t < PL::TPF(I)) {
#pragma unroll
            for (int g = 0; g < PL::G(I); ++g)
#pragma unroll
                for (int k = 1; k < PL::R(I); ++k)
                    tw[PL::tw_offset(I) + g * (PL::R(I) - 1) + k - 1] = twN[TS * mix_twiddle_index<PL, I>(t, g, k)];
        }
        plan_load_twiddles<PL, I + 1, TS>(t, twN, tw);
    }
}
// LDS table: pass I's block is [g][k - 1][t], so the threads of a wave read consecutive entries
template <class PL, int I>
__device__ __forceinline__ void plan_fill_table(int tid, const cf* __restrict__ twN, cf* table)
{
    if constexpr (I < PL::F - 1) {
        constexpr int R = PL::R(I), T = PL::TPF(I), n = PL::G(I) * (R - 1) * T;
        for (int i = tid; i < n; i += PL::WG) {
            const int t = i % T, gk = i / T, g = gk / (R - 1), k = gk % (R - 1) + 1;
            table[PL::tw_table_offset(I) + i] = twN[mix_twiddle_index<PL, I>(t, g, k)];
        }
        plan_fill_table<PL, I + 1>(tid, twN, table);
    }
}

// TW == 2: pass I >= 1's block is [k - 1][ntail]
template <class PL, int I, int TS = 1>
__device__ __forceinline__ void plan_fill_shared_table(int tid, const cf* __restrict__ twN, cf* table)
{
    if constexpr (I < PL::F - 1) {
        constexpr int S = PL::S(I), n = (PL::R(I) - 1) * S;
        for (int i = tid; i < n; i += PL::WG) table[PL::tw2_offset(I) + i] = twN[TS * (PL::D(I) * (i % S) * (i / S + 1))];
        plan_fill_shared_table<PL, I + 1, TS>(tid, twN, table);
    }
}

template <class PL, int I>
__device__ __forceinline__ void plan_later_passes(int t, cf* slab, const cf* tw, const cf* table, double* acc, bool active)
{
    if constexpr (I < PL::F) {
        constexpr int R = PL::R(I);
        constexpr bool last = I == PL::F - 1;
        if (PL::TPF(I) == PL::TPFMAX || t < PL::TPF(I)) {
#pragma unroll
            for (int g = 0; g < PL::G(I); ++g) {
                const int sb = mix_slot_base<PL, I>(t, g);
                cf v[R];
                mix_fetch<PL, I>(sb, v, slab);
                if constexpr (last) {
                    if constexpr (PL::WIDE) {
                        if (active) mix_last_pass_accumulate<PL>(v, acc + g * R);
                    } else {
                        mix_butterfly<PL, I>(v, nullptr);
                        if (active) phase_accumulate(v, acc + g * R, R);
                    }
                } else {
                    if constexpr (PL::TW == 0) {
                        mix_butterfly<PL, I>(v, tw + PL::tw_offset(I) + g * (R - 1));
                    } else {
                        cf twj[R - 1];
                        const cf* const row = PL::TW == 1 ? table + PL::tw_table_offset(I) + g * (R - 1) * PL::TPF(I) + t
                                                          : table + PL::tw2_offset(I) + (t + g * PL::TPF(I)) % PL::S(I);
#pragma unroll
                        for (int k = 0; k < R - 1; ++k) twj[k] = row[k * (PL::TW == 1 ? PL::TPF(I) : PL::S(I))];
                        mix_butterfly<PL, I>(v, twj);
                    }
                    mix_store<PL, I>(sb, v, slab);
                }
            }
        }
        if constexpr (!last) exchange_sync<true>();
        plan_later_passes<PL, I + 1>(t, slab, tw, table, acc, active);
    }
}

// pass 0 of butterflies G, G+1, ... of a thread: unpack from the raw registers, transform, store.
// (Measured and dropped: computing pass 0 into registers BEFORE the barrier that waits for the previous frame's
// last pass to leave the slab, storing after it -- 0 ... -5 % even where one workgroup has the CU to itself.)
template <class PL, bool WINDOW, int G>
__device__ __forceinline__ void plan_first_pass(int t, cf* slab, const cf* tw, const cf* table, const uint32_t* raw,
                                                const float* sgn, const float* wsgn)
{
    if constexpr (G < PL::G(0)) {
        constexpr int R0 = PL::R(0), T0 = PL::TPF(0);
        cf vg[R0];
        if constexpr (WINDOW && PL::WLDS) mix_unpack<PL, WINDOW, G * R0, PL::S(0)>(raw, sgn[G], wsgn + G * T0, vg);
        else mix_unpack<PL, WINDOW, G * R0>(raw, sgn[G], wsgn + (WINDOW ? G * R0 : 0), vg);
        if constexpr (PL::TW != 1) {
            mix_butterfly<PL, 0>(vg, tw + G * (R0 - 1));
        } else {
            cf twj[R0 - 1];
#pragma unroll
            for (int k = 0; k < R0 - 1; ++k) twj[k] = table[(G * (R0 - 1) + k) * T0 + t];
            mix_butterfly<PL, 0>(vg, twj);
        }
        mix_store<PL, 0>(mix_slot_base<PL, 0>(t, G), vg, slab);
        plan_first_pass<PL, WINDOW, G + 1>(t, slab, tw, table, raw, sgn, wsgn);
    }
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <class PL, bool WINDOW>
__global__ __launch_bounds__(PL::WG) void mixed_plan_kernel(const uint8_t* __restrict__ stream, long nframes,
                                                           const cf* __restrict__ twN, const float* __restrict__ window,
                                                           double* __restrict__ partial)
{
    constexpr int N = PL::N, R0 = PL::R(0), G0 = PL::G(0), T0 = PL::TPF(0), S0 = PL::S(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int fs = tid / PL::TPFMAX, t = tid - fs * PL::TPFMAX;
    cf* const slab = reinterpret_cast<cf*>(smem) + fs * PL::LDS_CPX;
    cf* const table = reinterpret_cast<cf*>(smem) + PL::FPW * PL::LDS_CPX;
    const bool in0 = T0 == PL::TPFMAX || t < T0;

    cf tw[PL::NTW_REG > 0 ? PL::NTW_REG : 1];
    if constexpr (PL::TW != 1) plan_load_twiddles<PL, 0>(t, twN, tw);
    if constexpr (PL::TW == 1) plan_fill_table<PL, 0>(tid, twN, table);
    if constexpr (PL::TW == 2) plan_fill_shared_table<PL, 1>(tid, twN, table);

    float sgn[G0];
    float wsgn[WINDOW && !PL::WLDS ? PL::PPT0 : 1];
    float* const wlds = reinterpret_cast<float*>(table + PL::TABLE_ENTRIES);   // window[n] (-1)^n
#pragma unroll
    for (int g = 0; g < G0; ++g) sgn[g] = ((t + g * T0) & 1) ? -1.0f : 1.0f;
    if constexpr (WINDOW && PL::WLDS) {
        for (int n = tid; n < N; n += PL::WG) wlds[n] = window[n] * ((n & 1) ? -1.0f : 1.0f);       // datastore.cxx:73,76-77
    } else if constexpr (WINDOW) {
#pragma unroll
        for (int g = 0; g < G0; ++g)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) {
                const int n = mix_sample_index<PL>(t, g, n1);
                wsgn[g * R0 + n1] = in0 ? window[n] * ((n & 1) ? -1.0f : 1.0f) : 0.0f;   // datastore.cxx:73,76-77
            }
    }
    double acc[PL::PPTL];
#pragma unroll
    for (int a = 0; a < PL::PPTL; ++a) acc[a] = 0.0;

    // the thread's samples of a frame: n = t + g T0 + n1 S0 (2-byte loads, consecutive across the wave),
    // two to a register (d16 / d16_hi loads)
    uint32_t raw[PL::NRAW];
    auto load_raw = [&](long frame) {
        const uint8_t* const p = stream + (frame < nframes ? frame : nframes - 1) * (2L * N) + 2 * t;
#pragma unroll
        for (int j = 0; j < PL::NRAW; ++j) {
            const int i0 = 2 * j, i1 = 2 * j + 1;
            us2 v;
            v.x = *reinterpret_cast<const uint16_t*>(p + 2 * ((i0 / R0) * T0 + (i0 % R0) * S0));
            v.y = i1 < PL::PPT0 ? *reinterpret_cast<const uint16_t*>(p + 2 * ((i1 / R0) * T0 + (i1 % R0) * S0)) : 0;
            raw[j] = __builtin_bit_cast(uint32_t, v);
        }
    };
    const long stride = static_cast<long>(gridDim.x) * PL::FPW;
    long fb = static_cast<long>(blockIdx.x) * PL::FPW;
    if (in0) load_raw(fb + fs);
    __syncthreads();
#pragma unroll 1
    for (; fb < nframes; fb += stride) {
        const bool active = (fb + fs) < nframes;
        exchange_sync<true>();                   // the previous frame's last pass has left the slab
        if (in0) {
            plan_first_pass<PL, WINDOW, 0>(t, slab, tw, table, raw, sgn, WINDOW && PL::WLDS ? wlds + t : wsgn);
            load_raw(fb + stride + fs);          // lands while the later passes run
        }
        exchange_sync<true>();
        plan_later_passes<PL, 1>(t, slab, tw, table, acc, active);
    }
    __syncthreads();
    // the slots' partial spectra through LDS into natural bin order, summed in slot order
    double* const stage = reinterpret_cast<double*>(smem);
    if (PL::TPF(PL::F - 1) == PL::TPFMAX || t < PL::TPF(PL::F - 1)) {
#pragma unroll
        for (int g = 0; g < PL::G(PL::F - 1); ++g)
#pragma unroll
            for (int k = 0; k < PL::RLAST; ++k) stage[fs * N + mix_bin<PL>(t, g, k)] = acc[g * PL::RLAST + k];
    }
    __syncthreads();
    for (int bin = tid; bin < N; bin += PL::WG) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < PL::FPW; ++k) v += stage[k * N + bin];
        partial[static_cast<size_t>(blockIdx.x) * N + bin] = v;
    }
}


// ---- split form: N = P M, M one of the planned lengths ------------------------------------------
// Workgroup b computes the residue p = b mod P of the spectrum, X[p + P k] = FFT_M(x'_p)[k] with
// x'_p[n] = (sum_j x[n + j M] W_P^{jp}) W_N^{np} (mixed_core.h, mix_unpack_split): the first radix-P
// pass of a decimation-in-frequency transform, evaluated for one output only, so nothing but the raw
// bytes ever crosses workgroups -- the P workgroups of a frame read the same 2N bytes (L2 / Infinity
// Cache serve the repeats) where the four-step kernels move 16 bytes of intermediate per sample.
// P = 2 ... 5 costs 5 P extra instructions per point.  Which sizes run this way is the table in
// mixed_plans_split.inc (20000 ... 80000 bins and 32768; large Bluestein served them at 0.04 Tsample/s).
// sections J, J + 1, ... of the frame: while section J is unpacked and added to v, section J + 1 is in flight
// into the other raw buffer (and, after the last one, section 0 of the workgroup's next frame)
template <class PL, int P, int WM, int J, class Load>
__device__ __forceinline__ void split_sections(uint32_t (*raw)[PL::PPT0], float (*wv)[WM == 2 ? PL::PPT0 : 1], const float* sgn,
                                               const float* w, const cf* wp, cf* v, const Load& load_next_frame_section0,
                                               const uint8_t* frame)
{
    if constexpr (J < P) {
        constexpr int T0 = PL::TPF(0), R0 = PL::R(0), S0 = PL::S(0);
        if constexpr (J + 1 < P) {
            const uint8_t* const base = frame + 2L * (J + 1) * PL::N;
#pragma unroll
            for (int i = 0; i < PL::PPT0; ++i)
                raw[(J + 1) & 1][i] = *reinterpret_cast<const uint16_t*>(base + 2 * ((i / R0) * T0 + (i % R0) * S0));
            if constexpr (WM == 2) {
#pragma unroll
                for (int i = 0; i < PL::PPT0; ++i) wv[(J + 1) & 1][i] = w[(J + 1) * PL::N + (i / R0) * T0 + (i % R0) * S0];
            }
        } else {
            load_next_frame_section0(raw[(J + 1) & 1], wv[(J + 1) & 1]);
        }
        mix_split_accumulate<PL, WM, J == 0>(raw[J & 1], sgn, WM == 2 ? wv[J & 1] : WM != 0 ? w + J * PL::N : w, wp[J], v);
        split_sections<PL, P, WM, J + 1>(raw, wv, sgn, w, wp, v, load_next_frame_section0, frame);
    }
}

// ROLL: the same pipeline on ONE buffer -- sample i of the section is consumed and its register (and its window
// value's, WM == 2) refilled at once with sample i of the NEXT section (after the last one: section 0 of the
// workgroup's next frame), so each load has a whole section's arithmetic to land and the pipeline costs PPT0
// registers instead of 2 PPT0 (+ PPT0 instead of 2 PPT0 for the window values).
//   nraw / nw: the next section's samples / window values at this thread's ntail (nw: global memory)
template <class PL, int WM, bool FIRST, int I = 0>
__device__ __forceinline__ void split_section_rolling(uint32_t* raw, float* wv, const float* sgn, const float* wl, cf wpj, cf* v,
                                                      const uint8_t* nraw, const float* nw)
{
    if constexpr (I < PL::PPT0) {
        constexpr int T0 = PL::TPF(0), R0 = PL::R(0), S0 = PL::S(0);
        constexpr int off = (I / R0) * T0 + (I % R0) * S0;
        float wi = 0.0f;
        if constexpr (WM == 2) wi = wv[I];
        else if constexpr (WM == 3) wi = wl[off];
        mix_split_element<PL, WM, FIRST, I>(raw[I], sgn, wi, wpj, v);
        raw[I] = *reinterpret_cast<const uint16_t*>(nraw + 2 * off);
        if constexpr (WM == 2) wv[I] = nw[off];
        split_section_rolling<PL, WM, FIRST, I + 1>(raw, wv, sgn, wl, wpj, v, nraw, nw);
    }
}
template <class PL, int P, int WM, int J = 0>
__device__ __forceinline__ void split_sections_rolling(uint32_t* raw, float* wv, const float* sgn, const float* wl, const float* w,
                                                       const cf* wp, cf* v, const uint8_t* frame, const uint8_t* next_frame)
{
    if constexpr (J < P) {
        const uint8_t* const nraw = J + 1 < P ? frame + 2L * (J + 1) * PL::N : next_frame;
        const float* const nw = J + 1 < P ? w + (J + 1) * PL::N : w;
        split_section_rolling<PL, WM, J == 0>(raw, wv, sgn, WM == 3 ? wl + J * PL::N : wl, wp[J], v, nraw, nw);
        split_sections_rolling<PL, P, WM, J + 1>(raw, wv, sgn, wl, w, wp, v, frame, next_frame);
    }
}

// Paired form (P = 2 Q = 6, 8, 10; mix_split_pair_element): sections j and j + Q ride the one-buffer pipeline side by
// side -- ra / rb (and wa / wb) hold pair j and are refilled, sample by sample, with pair j + 1 (after the last one:
// pair 0 of the workgroup's next frame).
template <class PL, int Q, int WM, bool FIRST, int I = 0>
__device__ __forceinline__ void split_pair_rolling(uint32_t* ra, uint32_t* rb, float* wa, float* wb, const float* sgn, float sp,
                                                   cf wpj, cf* v, const uint8_t* na, const float* nw)
{
    if constexpr (I < PL::PPT0) {
        constexpr int T0 = PL::TPF(0), R0 = PL::R(0), S0 = PL::S(0);
        constexpr int off = (I / R0) * T0 + (I % R0) * S0;
        mix_split_pair_element<PL, WM, FIRST, I>(ra[I], rb[I], sgn, sp, WM == 2 ? wa[I] : 0.0f, WM == 2 ? wb[I] : 0.0f, wpj, v);
        ra[I] = *reinterpret_cast<const uint16_t*>(na + 2 * off);
        rb[I] = *reinterpret_cast<const uint16_t*>(na + 2L * Q * PL::N + 2 * off);
        if constexpr (WM == 2) {
            wa[I] = nw[off];
            wb[I] = nw[Q * PL::N + off];
        }
        split_pair_rolling<PL, Q, WM, FIRST, I + 1>(ra, rb, wa, wb, sgn, sp, wpj, v, na, nw);
    }
}
template <class PL, int Q, int WM, int J = 0>
__device__ __forceinline__ void split_pairs_rolling(uint32_t* ra, uint32_t* rb, float* wa, float* wb, const float* sgn, float sp,
                                                    const float* w, const cf* wp, cf* v, const uint8_t* frame,
                                                    const uint8_t* next_frame)
{
    if constexpr (J < Q) {
        const uint8_t* const na = J + 1 < Q ? frame + 2L * (J + 1) * PL::N : next_frame;
        const float* const nw = J + 1 < Q ? w + (J + 1) * PL::N : w;
        split_pair_rolling<PL, Q, WM, J == 0>(ra, rb, wa, wb, sgn, sp, wp[J], v, na, nw);
        split_pairs_rolling<PL, Q, WM, J + 1>(ra, rb, wa, wb, sgn, sp, w, wp, v, frame, next_frame);
    }
}

// WM: how a windowed run gets its window values (mix_split_accumulate): 1 = loaded where they are used (every
// section waits for its loads: one workgroup per CU has nothing else to run meanwhile); 2 = fetched one section
// ahead, with the raw samples (PPT0 more registers); 3 = window[n] (-1)^n for the whole frame in LDS, filled once
// per workgroup (4 N bytes beside the M-point slab).
template <class PL, int P, int WM, bool ROLL = false>
__global__ __launch_bounds__(PL::WG) void mixed_split_kernel(const uint8_t* __restrict__ stream, long nframes,
                                                            const cf* __restrict__ twN, const float* __restrict__ window,
                                                            double* __restrict__ partial)
{
    static_assert(PL::TW != 1 && PL::FPW == 1 && PL::N % 2 == 0, "pass-0 twiddles in registers, one frame slot, even M");
    constexpr int M = PL::N, N = P * M, R0 = PL::R(0), G0 = PL::G(0), T0 = PL::TPF(0), S0 = PL::S(0);
    static_assert(WM != 3 || PL::LDS_BYTES + 4 * N <= 160 * 1024, "no room for the window in LDS");
    static_assert(!ROLL || WM != 1, "the rolling pipeline fetches its window values ahead (2) or keeps them in LDS (3)");
    constexpr bool PAIR = P > 5;             // P = 6, 8, 10: the paired form (split_pairs_rolling), Q = P / 2 terms
    constexpr int Q = PAIR ? P / 2 : P;
    static_assert(!PAIR || (P % 2 == 0 && (WM == 0 || WM == 2)), "paired form: even P, window values fetched ahead");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    // The P workgroups of a frame read the same bytes: put them on one XCD (workgroups go to the 8 XCDs round-robin by
    // index), so that its L2 fetches the frame once and serves the others -- when the grid is a multiple of 8 P;
    // any grid that is a multiple of P works with the plain mapping.
    const bool xcd_local = gridDim.x % (8 * P) == 0;
    const int q = blockIdx.x / 8;
    const int p = xcd_local ? q % P : blockIdx.x % P;
    const int group = xcd_local ? (blockIdx.x % 8) + 8 * (q / P) : blockIdx.x / P;
    cf* const slab = reinterpret_cast<cf*>(smem);
    cf* const table = reinterpret_cast<cf*>(smem) + PL::LDS_CPX;
    float* const wlds = reinterpret_cast<float*>(table + PL::TABLE_ENTRIES);      // WM == 3
    const bool in0 = T0 == PL::TPFMAX || t < T0;

    cf tw[PL::NTW_REG > 0 ? PL::NTW_REG : 1];                    // (pass 0's block is unused here)
    plan_load_twiddles<PL, 0, P>(t, twN, tw);
    if constexpr (PL::TW == 2) plan_fill_shared_table<PL, 1, P>(t, twN, table);
    if constexpr (WM == 3) {
        for (int n = t; n < N; n += PL::WG) wlds[n] = window[n] * ((n & 1) ? -1.0f : 1.0f);        // datastore.cxx:73,76-77
    }
    // pass 0: output k of the butterfly with ntail carries W_N^{ntail p} W_M^{ntail k} = W_N^{ntail (p + P k)}
    cf tw0[G0 * R0];
    float sgn[G0];
#pragma unroll
    for (int g = 0; g < G0; ++g) {
        const int ntail = t + g * T0;
        sgn[g] = (ntail & 1) ? -1.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < R0; ++k) tw0[g * R0 + k] = in0 ? twN[(static_cast<long>(ntail) * (p + P * k)) % N] : cf{0.0f, 0.0f};
    }
    cf wp[Q], mid[R0];                                           // the same for every thread of the workgroup
#pragma unroll
    for (int j = 0; j < Q; ++j) wp[j] = twN[(static_cast<long>(j) * p * M) % N];
    const float sp = (p & 1) ? -1.0f : 1.0f;                     // (paired form)
#pragma unroll
    for (int n1 = 0; n1 < R0; ++n1) mid[n1] = twN[(static_cast<long>(n1) * S0 * p) % N];
    double acc[PL::PPTL];
#pragma unroll
    for (int a = 0; a < PL::PPTL; ++a) acc[a] = 0.0;

    uint32_t raw[ROLL && !PAIR ? 1 : 2][PL::PPT0];          // one sample per register (mix_split_accumulate)
    float wv[ROLL && !PAIR ? 1 : 2][WM == 2 ? PL::PPT0 : 1];
    auto load_section = [&](long frame, int j, uint32_t* dst, float* wdst) {
        const uint8_t* const base = stream + (frame < nframes ? frame : nframes - 1) * (2L * N) + 2L * j * M + 2 * t;
#pragma unroll
        for (int i = 0; i < PL::PPT0; ++i) dst[i] = *reinterpret_cast<const uint16_t*>(base + 2 * ((i / R0) * T0 + (i % R0) * S0));
        if constexpr (WM == 2) {
            // (opaque per frame: otherwise the compiler keeps section 0's values in registers across the loop)
            const float* w0 = window + j * M + t;
            asm volatile("" : "+v"(w0));
#pragma unroll
            for (int i = 0; i < PL::PPT0; ++i) wdst[i] = w0[(i / R0) * T0 + (i % R0) * S0];
        }
    };
    auto load_section0 = [&](long frame, uint32_t* dst, float* wdst) { load_section(frame, 0, dst, wdst); };
    const long stride = gridDim.x / P;
    long fb = group;
    // two buffers: section j of a frame lives in raw[j & 1]; the last section's turn puts the next frame's section 0
    // into raw[P & 1] -- for odd P that is raw[1], moved to raw[0] at the top of the next frame (long landed by then)
    constexpr int B0 = ROLL || PAIR ? 0 : (P & 1);
    if (in0) {
        load_section0(fb, raw[B0], wv[B0]);
        if constexpr (PAIR) load_section(fb, Q, raw[1], wv[1]);
    }
    __syncthreads();
#pragma unroll 1
    for (; fb < nframes; fb += stride) {
        cf v[PL::PPT0];
        if (in0) {
            const uint8_t* const frame = stream + fb * (2L * N) + 2 * t;
            // (opaque per frame: otherwise the compiler hoists the loop-invariant window loads out of the frame
            // loop and spills them)
            const float* w = WM == 3 ? wlds + t : window + t;
            if constexpr (WM == 1 || WM == 2) asm volatile("" : "+v"(w));
            const long next = fb + stride;
            if constexpr (PAIR) {
                const uint8_t* const next_frame = stream + (next < nframes ? next : nframes - 1) * (2L * N) + 2 * t;
                split_pairs_rolling<PL, Q, WM>(raw[0], raw[1], wv[0], wv[1], sgn, sp, w, wp, v, frame, next_frame);
            } else if constexpr (ROLL) {
                const uint8_t* const next_frame = stream + (next < nframes ? next : nframes - 1) * (2L * N) + 2 * t;
                split_sections_rolling<PL, P, WM>(raw[0], wv[0], sgn, w, w, wp, v, frame, next_frame);
            } else {
                if constexpr (P % 2 == 1) {
#pragma unroll
                    for (int r = 0; r < PL::PPT0; ++r) raw[0][r] = raw[B0][r];
                    if constexpr (WM == 2) {
#pragma unroll
                        for (int r = 0; r < PL::PPT0; ++r) wv[0][r] = wv[B0][r];
                    }
                }
                auto next0 = [&](uint32_t* dst, float* wdst) { load_section0(next, dst, wdst); };
                split_sections<PL, P, WM, 0>(raw, wv, sgn, w, wp, v, next0, frame);
            }
            mix_split_mid<PL>(v, mid);
        }
        exchange_sync<true>();                   // the previous frame's last pass has left the slab
        if (in0) {
#pragma unroll
            for (int g = 0; g < G0; ++g) {
                mix_butterfly_split<PL>(v + g * R0, tw0 + g * R0);
                mix_store<PL, 0>(mix_slot_base<PL, 0>(t, g), v + g * R0, slab);
            }
        }
        exchange_sync<true>();
        plan_later_passes<PL, 1>(t, slab, tw, table, acc, true);
    }
    __syncthreads();
    // the residue's M bins through LDS into natural order, then into the partial spectrum this workgroup shares
    // with the P - 1 others of its group (disjoint bins)
    double* const stage = reinterpret_cast<double*>(smem);
    if (PL::TPF(PL::F - 1) == PL::TPFMAX || t < PL::TPF(PL::F - 1)) {
#pragma unroll
        for (int g = 0; g < PL::G(PL::F - 1); ++g)
#pragma unroll
            for (int k = 0; k < PL::RLAST; ++k) stage[mix_bin<PL>(t, g, k)] = acc[g * PL::RLAST + k];
    }
    __syncthreads();
    double* const row = partial + static_cast<size_t>(group) * N + p;
    for (int k = t; k < M; k += PL::WG) row[static_cast<size_t>(P) * k] = stage[k];
}

template <class PL>
constexpr int plan_lds_bytes(bool windowed)
{
    return PL::LDS_BYTES +
           (windowed && PL::WLDS ? PL::N * (int)sizeof(float) : 0);
}
template <class PL, bool WINDOW>
constexpr PlanForm plan_form()
{
    return {mixed_plan_kernel<PL, WINDOW>, PL::WG, PL::FPW, plan_lds_bytes<PL>(WINDOW), 1};
}
// the split form's window mode when the table does not name one: the LDS copy where it fits, else loads in place
template <int P, class PL>
constexpr int split_window_mode()
{
    return PL::LDS_BYTES + 4 * P * PL::N <= 160 * 1024 ? 3 : 1;
}
template <int P, class PL, int WM, bool ROLL = false>
constexpr PlanForm split_form()
{
    return {mixed_split_kernel<SplitPlan<P, PL>, P, WM, ROLL>, PL::WG, 1, PL::LDS_BYTES + (WM == 3 ? 4 * P * PL::N : 0), P};
}
template <class PL>
constexpr PlanEntry plan_entry(int variant)
{
    return {PL::N, variant, plan_form<PL, false>(), plan_form<PL, true>()};
}
// (WM: window mode of the windowed twin, mixed_split_kernel; 0 = split_window_mode's choice.  ROLL: the one-buffer
// section pipeline, for the plans that have no registers for two)
template <int P, class PL, int WM = 0, bool ROLL = false>
constexpr PlanEntry split_entry(int variant)
{
    constexpr int wm = P > 5 ? 2 : WM ? WM : split_window_mode<P, PL>();       // (the paired form fetches its window values ahead)
    return {P * PL::N, variant, split_form<P, PL, 0, ROLL>(), split_form<P, PL, (ROLL && wm == 1) ? 2 : wm, ROLL>()};
}
template <int R, int G = 1>
using P = MPass<R, G>;

#ifdef RPF_TUNING
// a candidate of the planned kernel (tools/gen_mixed_plans.py search), without its windowed twin
template <class PL>
constexpr PlanEntry plan_candidate(int variant)
{
    return {PL::N, variant, plan_form<PL, false>(), PlanForm{nullptr, 0, 0, 0, 1}};
}
#endif

}  // namespace

}  // namespace rpf
