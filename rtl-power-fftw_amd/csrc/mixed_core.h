// mixed_core.h -- per-thread building blocks of KM, the LDS-resident mixed-radix kernel
// (rpf_mixed.hip): K1's scheme (fft_core.h) carried over to lengths N = R_0 R_1 ... R_{F-1} whose
// radices are the small composites of dft_small.h (2 ... 25).
//
// Decimation in frequency, in place by element name e in [0, N) (= the time index n in pass 0):
// with S_i = R_{i+1} ... R_{F-1} and D_i = R_0 ... R_{i-1}, pass i runs N / R_i butterflies
//
//   b = khead S_i + ntail   (khead < D_i, ntail < S_i)    elements  e_n = khead R_i S_i + n S_i + ntail
//   v <- DFT_{R_i}(v);   v[k] *= W_N^{D_i ntail k}   (no twiddles in the last pass: S = 1)
//
// and the value that ends in element e = sum_i k_i S_i is X[sum_i k_i D_i] (digit reversal in the
// mixed radix system; free, because the only consumer is a per-bin accumulator that lives
// wherever its bin lands).  A thread owns G_i butterflies of pass i -- b = t + g TPF_i,
// TPF_i = N / (R_i G_i) threads of the frame take part in pass i -- and the SAME butterflies in
// every frame, so its twiddles are loop-invariant registers (or a per-thread row of an LDS table)
// and its accumulators never move.  Between passes the frame goes through one padded LDS slab:
// slot(e) = e + e / R_last keeps the last pass's stride-R_last fetch off the same banks, and
// because every S_i (i < F-1) is a multiple of R_last the slot of e_n is slot(e_0) plus a
// compile-time constant -- the DS instruction's immediate offset.
// Plain C++17 on registers and pointers: compiled by hipcc into the gfx950 kernel and by the host
// emulator under tests/emul that checks the index maps thread by thread without a GPU.
#pragma once

#include "dft_small.h"
#include "dft_small_wide.h"

namespace rpf {

template <int R_, int G_ = 1>
struct MPass {
    static constexpr int R = R_, G = G_;
};

// TW: where a thread's twiddles live -- 0 registers; 1 per-thread rows of an LDS table; 2 registers for pass 0
// (every thread's are different) and, for the later passes, one LDS table per pass indexed by [k][ntail] (threads
// with the same ntail share an entry: (R_i - 1) S_i entries, a few percent of N).
// TW + 4 (kWideLast): the LAST pass runs in double (dft_small_wide.h, mix_last_pass_accumulate) -- the pass in which a
// float32 transform of a tone-rich frame loses its accuracy; the split / paired forms' plans carry it.
// TW + 12 (kWideLast | kWideLastTwo): so does the pass before it (mix_butterfly: double butterfly and twiddle products,
// one rounding to float where the values go back to the slab) -- the forms of 60000 bins and more.
constexpr int kWideLast = 4, kWideLastTwo = 8;
template <int N_, int FPW_, int TW_, class... Ps>
struct MixPlan {
    static constexpr int N = N_, FPW = FPW_, TW = TW_ & 3;
    static constexpr bool WIDE = (TW_ & kWideLast) != 0;
    static constexpr bool WIDE2 = (TW_ & kWideLastTwo) != 0;
    static constexpr int F = sizeof...(Ps);
    static constexpr int Rs[F] = {Ps::R...};
    static constexpr int Gs[F] = {Ps::G...};
    static constexpr int R(int i) { return Rs[i]; }
    static constexpr int G(int i) { return Gs[i]; }
    static constexpr int S(int i)        // product of the later radices
    {
        int s = 1;
        for (int j = i + 1; j < F; ++j) s *= Rs[j];
        return s;
    }
    static constexpr int D(int i)        // product of the earlier radices
    {
        int d = 1;
        for (int j = 0; j < i; ++j) d *= Rs[j];
        return d;
    }
    static constexpr int TPF(int i) { return N / (Rs[i] * Gs[i]); }
    static constexpr int tpf_max()
    {
        int m = 0;
        for (int i = 0; i < F; ++i) m = TPF(i) > m ? TPF(i) : m;
        return m;
    }
    static constexpr int TPFMAX = tpf_max();
    static constexpr int WG = FPW * TPFMAX;
    static constexpr int RLAST = Rs[F - 1];
    static constexpr bool PAD = (RLAST % 2) == 0;      // an odd stride is conflict-free as it is
    static RPF_HD int slot(int e) { return PAD ? e + e / RLAST : e; }
    static constexpr int LDS_CPX = PAD ? N + N / RLAST : N;
    static constexpr int tw_offset(int i)     // first twiddle of pass i in a thread's list
    {
        int o = 0;
        for (int j = 0; j < i; ++j) o += Gs[j] * (Rs[j] - 1);
        return o;
    }
    static constexpr int NTW = tw_offset(F - 1);
    static constexpr int tw_table_offset(int i)   // first entry of pass i's block of the LDS table
    {
        int o = 0;
        for (int j = 0; j < i; ++j) o += Gs[j] * (Rs[j] - 1) * TPF(j);
        return o;
    }
    static constexpr int TW_TABLE = tw_table_offset(F - 1);
    static constexpr int tw2_offset(int i)        // TW == 2: first entry of pass i >= 1
    {
        int o = 0;
        for (int j = 1; j < i; ++j) o += (Rs[j] - 1) * S(j);
        return o;
    }
    static constexpr int TW2_TABLE = tw2_offset(F - 1);
    static constexpr int NTW_REG = TW == 0 ? NTW : TW == 2 ? Gs[0] * (Rs[0] - 1) : 0;        // twiddle registers
    static constexpr int TABLE_ENTRIES = TW == 1 ? TW_TABLE : TW == 2 ? TW2_TABLE : 0;       // twiddles in LDS
    static constexpr int PPT0 = Rs[0] * Gs[0];
    static constexpr int NRAW = (PPT0 + 1) / 2;          // raw registers: two samples each
    static constexpr int LDS_BYTES = (FPW * LDS_CPX + TABLE_ENTRIES) * 8;
    // window values: the thread's registers, or (room permitting) LDS when it has many of them or few registers
    static constexpr bool WLDS = LDS_BYTES + 4 * N <= 160 * 1024;
    static constexpr int PPTL = Rs[F - 1] * Gs[F - 1];
    static constexpr bool valid()
    {
        int p = 1;
        for (int i = 0; i < F; ++i) {
            p *= Rs[i];
            if (N % (Rs[i] * Gs[i]) != 0) return false;
        }
        return p == N && F >= 2;
    }
    static_assert(valid(), "radices must multiply to N and R_i G_i must divide N");
};

// The same plan with the wide last pass (kWideLast).
template <class PL>
struct WidePlanOf;
template <int N_, int FPW_, int TW_, class... Ps>
struct WidePlanOf<MixPlan<N_, FPW_, TW_, Ps...>> {
    using type = MixPlan<N_, FPW_, TW_ | kWideLast, Ps...>;
    using type2 = MixPlan<N_, FPW_, TW_ | kWideLast | kWideLastTwo, Ps...>;
};
template <class PL>
using WidePlan = typename WidePlanOf<PL>::type;
template <class PL>
using WidePlan2 = typename WidePlanOf<PL>::type2;

// The split / paired forms of 40000 bins and more run their M-point plan with the WIDE last pass (mixed_core.h: the last
// butterfly and the squares in double): from there on a float32 last pass beside a strong line leaves less than 15 % of
// the parity bar against the CPU path on at least one of the recorded tone streams (profiles/r04_fullsize_errors.json:
// 8.3e-7 ... 9.96e-7 from 42000 bins up), and so do four smaller sizes (8.0 ... 8.6e-7).  Below, the float pass keeps
// >= 20 % and its speed (the wide pass costs 7 % in the median and up to 2 x where its 4 R_last registers spill).
// DESIGN.md 6; RPF_SPLIT_WIDE=0 (make f32pass): every form on the float pass, for A/B.
#ifndef RPF_SPLIT_WIDE
#define RPF_SPLIT_WIDE 1
#endif
constexpr bool split_is_wide(int n)
{
    return RPF_SPLIT_WIDE != 0 && (n >= 40000 || n == 21000 || n == 32000 || n == 34000 || n == 35000);
}
// ... and the pass before it too where the last radix is small.  What a pass costs the weak bins beside a line falls with
// the line's concentration at that pass: ~ kappa eps A / sqrt(S_i) on (R_i - 1) S_i bins, S_i = the product of the later
// radices (tools/analysis/parity_passes.cpp).  Behind a last radix of 20 ... 25 the pass before the last is a fifth of
// what the last pass was and float32 is good enough (measured, last pass wide: 2.8 - 4.5e-7 from the truth at 66000,
// 88000, 92000); behind a last radix of 8 ... 16 it is a quarter to a third, and those forms stayed 0.7 - 0.8e-6 from
// the truth on their worst held-out stream (75000, 81000, 90000, 108000: all four plans end in 10 ... 15).
template <class PL>
constexpr bool split_is_wide2(int n) { return split_is_wide(n) && PL::RLAST < 20; }
template <int P, class PL>
using SplitPlan = std::conditional_t<split_is_wide2<PL>(P * PL::N), WidePlan2<PL>, std::conditional_t<split_is_wide(P * PL::N), WidePlan<PL>, PL>>;

// slot of register 0 of butterfly g of thread t in pass I, and the constant added for register n
template <class PL, int I>
RPF_HD int mix_slot_base(int t, int g)
{
    constexpr int S = PL::S(I), R = PL::R(I);
    const int b = t + g * PL::TPF(I);
    const int khead = b / S, ntail = b - khead * S;
    return PL::slot(khead * R * S + ntail);
}
template <class PL, int I>
constexpr int mix_slot_delta(int n)
{
    return n * PL::S(I) + (PL::PAD ? (n * PL::S(I)) / PL::RLAST : 0);
}
// index into the master table W_N^k of the twiddle of output k of butterfly g (pass I < F-1)
template <class PL, int I>
RPF_HD int mix_twiddle_index(int t, int g, int k)
{
    const int b = t + g * PL::TPF(I);
    return PL::D(I) * (b % PL::S(I)) * k;
}
// time index of input n1 of butterfly g of pass 0
template <class PL>
RPF_HD int mix_sample_index(int t, int g, int n1)
{
    return n1 * PL::S(0) + t + g * PL::TPF(0);
}
// spectrum bin of output k of butterfly g of the last pass
template <class PL, int I = PL::F - 2>
RPF_HD int mix_bin_head(int khead)
{
    // khead = ((k_0 R_1 + k_1) R_2 + ...) + k_{F-2}: peel the digits from the least significant
    if constexpr (I >= 0) {
        const int q = khead / PL::R(I), k = khead - q * PL::R(I);
        return k * PL::D(I) + mix_bin_head<PL, I - 1>(q);
    } else {
        return 0;
    }
}
template <class PL>
RPF_HD int mix_bin(int t, int g, int k)
{
    return mix_bin_head<PL>(t + g * PL::TPF(PL::F - 1)) + k * PL::D(PL::F - 1);
}

// Pass 0, butterfly g: unpack R_0 samples (datastore.cxx:73-77) from the thread's raw registers,
// two 16-bit samples per register (sample n1 of butterfly g = half (g R_0 + n1) & 1 of register
// (g R_0 + n1) / 2).  sgn = (-1)^(t + g TPF_0); wsgn[n1] = window * (-1)^n of the same samples.
// WSTRIDE: distance between the window values of consecutive n1 (1: the thread's registers; S_0: the
// workgroup's LDS copy of window[n] (-1)^n, used when the registers are needed elsewhere).
template <class PL, bool WINDOW, int GOFF, int WSTRIDE = 1, int N1 = 0>
RPF_HD void mix_unpack(const uint32_t* raw, float sgn, const float* wsgn, cf* v)
{
    if constexpr (N1 < PL::R(0)) {
        constexpr int idx = GOFF + N1;
        const cf f = iq_pair_plus_2p23<idx & 1>(raw[idx >> 1]);
        if constexpr (WINDOW) {
            v[N1] = (f - (kTwo23 + 127.0f)) * wsgn[N1 * WSTRIDE];      // (v - 127) exact, one rounding
        } else {
            const float sg = ((N1 * PL::S(0)) & 1) ? -sgn : sgn;       // (-1)^(n1 S_0 + ntail)
            v[N1] = f * sg - (kTwo23 + 127.0f) * sg;                   // every step exact
        }
        mix_unpack<PL, WINDOW, GOFF, WSTRIDE, N1 + 1>(raw, sgn, wsgn, v);
    }
}

// Split form (N = P M, one workgroup per residue p: X[p + P k] = FFT_M(x'_p)[k]): pass-0 input
//   x'_p[n] = ( sum_{j<P} x[n + j M] W_P^{j p} ) W_N^{n p},     n = ntail + n1 S_0,
//   W_N^{n p} = W_N^{ntail p} (folded into the butterfly's output twiddles) * W_N^{n1 S_0 p} (mid[n1], the same for
//   every thread of the workgroup).
// The sum runs section by section (j = 0, 1, ...: the j-th M samples of the frame), so that only one section's
// raw registers (and the next one's, in flight) are live whatever P is:
//   mix_split_accumulate<J == 0>: v[i] (+)= x_j[i] W_P^{j p} for the thread's PPT0 samples i = g R_0 + n1
//     raw: the section's samples, one per register (the two buffers of the pipeline are loaded and consumed
//     without a packing step in between, which would wait for the loads where they are issued);
//     w: its window values (see WM below); wpj = W_P^{j p}
//   mix_split_mid: v[g R_0 + n1] *= mid[n1]
// M is even, so (-1)^(n + j M) does not depend on j.
// WM: where the window values of the section come from -- 0: no window; 1: w[ntail + n1 S_0] (global memory,
// plain values); 2: w[i], the thread's PPT0 values in registers (plain, fetched a section ahead like the raw
// samples); 3: w[ntail + n1 S_0] in the workgroup's LDS copy of window[n] (-1)^n (sign included).
// one sample of the section: raw_i = its 16-bit IQ pair, wi = its window value (WM 1, 2: plain; 3: sign included)
template <class PL, int WM, bool FIRST, int I>
RPF_HD void mix_split_element(uint32_t raw_i, const float* sgn, float wi, cf wpj, cf* v)
{
    constexpr int g = I / PL::R(0), n1 = I % PL::R(0);
    const float sg = ((n1 * PL::S(0)) & 1) ? -sgn[g] : sgn[g];
    const cf f = iq_plus_2p23(raw_i);
    cf x;
    if constexpr (WM == 1 || WM == 2) x = (f - (kTwo23 + 127.0f)) * (wi * sg);      // one rounding
    else if constexpr (WM == 3) x = (f - (kTwo23 + 127.0f)) * wi;
    else x = f * sg - (kTwo23 + 127.0f) * sg;                                       // exact
    if constexpr (FIRST) v[I] = x;
    else v[I] = cmac_k(v[I], x, wpj);           // (wpj, mid: the same in every lane -- scalar registers)
}
// Paired form (P = 2 Q > 5): W_P^{(j+Q) p} = (-1)^p W_P^{j p}, so sections j and j + Q are added first --
// exactly, the samples being small integers (a windowed pair costs one rounding more) -- and the sum has Q terms:
// the first pass of a P = 10 split is as accurate as P = 5's.  sp = (-1)^p; wa, wb: the two samples' window values
// (WM == 2: plain; WM == 0: unused).
template <class PL, int WM, bool FIRST, int I>
RPF_HD void mix_split_pair_element(uint32_t raw_a, uint32_t raw_b, const float* sgn, float sp, float wa, float wb, cf wpj, cf* v)
{
    static_assert(WM == 0 || WM == 2, "paired form: no window, or window values fetched ahead");
    constexpr int g = I / PL::R(0), n1 = I % PL::R(0);
    const float sg = ((n1 * PL::S(0)) & 1) ? -sgn[g] : sgn[g];
    const cf a = iq_plus_2p23(raw_a) - (kTwo23 + 127.0f);        // exact
    const cf b = iq_plus_2p23(raw_b) - (kTwo23 + 127.0f);
    cf x;
    if constexpr (WM == 2) x = a * (wa * sg) + b * (wb * (sg * sp));
    else x = (a + b * sp) * sg;                                  // exact: |.| <= 256
    if constexpr (FIRST) v[I] = x;
    else v[I] = cmac_k(v[I], x, wpj);
}
template <class PL, int WM, bool FIRST, int I = 0>
RPF_HD void mix_split_accumulate(const uint32_t* raw, const float* sgn, const float* w, cf wpj, cf* v)
{
    if constexpr (I < PL::PPT0) {
        constexpr int g = I / PL::R(0), n1 = I % PL::R(0);
        float wi = 0.0f;
        if constexpr (WM == 2) wi = w[I];
        else if constexpr (WM != 0) wi = w[g * PL::TPF(0) + n1 * PL::S(0)];
        mix_split_element<PL, WM, FIRST, I>(raw[I], sgn, wi, wpj, v);
        mix_split_accumulate<PL, WM, FIRST, I + 1>(raw, sgn, w, wpj, v);
    }
}
template <class PL, int I = 0>
RPF_HD void mix_split_mid(cf* v, const cf* mid)
{
    if constexpr (I < PL::PPT0) {
        v[I] = cmul_k(v[I], mid[I % PL::R(0)]);
        mix_split_mid<PL, I + 1>(v, mid);
    }
}

// pass-0 butterfly of the split form: every output, k = 0 included, carries the factor W_N^{ntail p}:
// tw[0] = that factor, tw[k] = factor * W_M^{ntail k}
template <class PL>
RPF_HD void mix_butterfly_split(cf* v, const cf* tw)
{
    constexpr int R = PL::R(0);
    SmallDft<R>::run(v);
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = cmul(v[k], tw[k]);
}

template <class PL, int I>
RPF_HD void mix_butterfly(cf* v, const cf* tw)
{
    constexpr int R = PL::R(I);
    if constexpr (PL::WIDE2 && I == PL::F - 2 && I >= 1) {
        // the pass before the last, wide: butterfly and twiddle products in double (the twiddles are the float table's),
        // one rounding where the values go back to the slab
        cf o[R];
        WidePass<R>::run(v, WideTwiddleStore{o, tw});
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = o[k];
    } else {
        SmallDft<R>::run(v);
        if constexpr (I < PL::F - 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = cmul(v[k], tw[k - 1]);
        }
    }
}

// The last pass of butterfly v (R_last values fetched from the slab) and pwr += |X|^2 (datastore.cxx:83-85) into the
// butterfly's R_last accumulators.  Float plans: the float butterfly, squares folded in as in phase_accumulate.
// Wide plans: the values go to double first, the butterfly and the squares are double arithmetic.
template <class PL>
RPF_HD void mix_last_pass_accumulate(cf* v, double* acc)
{
    constexpr int R = PL::RLAST;
    if constexpr (PL::WIDE) {
        WidePass<R>::run(v, WideFold{acc});
    } else {
        mix_butterfly<PL, PL::F - 1>(v, nullptr);
        phase_accumulate(v, acc, R);
    }
}

template <class PL, int I>
RPF_HD void mix_store(int slot_base, const cf* v, cf* slab)
{
    cf* const p = slab + slot_base;
#pragma unroll
    for (int n = 0; n < PL::R(I); ++n) p[mix_slot_delta<PL, I>(n)] = v[n];
}
template <class PL, int I>
RPF_HD void mix_fetch(int slot_base, cf* v, const cf* slab)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // (volatile LDS pointer: keeps hipcc from pairing the loads into the half-rate ds_read2_b64)
    using lds_cf = const volatile __attribute__((address_space(3))) cf;
    lds_cf* const p = (lds_cf*)(slab + slot_base);
#else
    const cf* const p = slab + slot_base;
#endif
#pragma unroll
    for (int n = 0; n < PL::R(I); ++n) v[n] = p[mix_slot_delta<PL, I>(n)];
}

}  // namespace rpf
