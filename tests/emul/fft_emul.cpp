// fft_emul.cpp -- TEST-ONLY thread-by-thread emulator of the fused gfx950
// kernel (rtl-power-fftw_amd/csrc/rpf_kernels.hip).  It runs the very same
// per-thread phase functions (fft_core.h) on the host, one "thread" after the
// other with a plain array standing in for LDS and a full barrier between
// phases, so the index maps (elem_of, slot, twiddle_index, bin_of), the
// butterflies and the unpack arithmetic can be checked against the oracle on a
// machine without a GPU.  It is not a product path and nothing in
// rtl-power-fftw_amd/ links it.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../rtl-power-fftw_amd/csrc/bluestein_tables.h"
#include "../../rtl-power-fftw_amd/csrc/fft_core.h"
#include "../../rtl-power-fftw_amd/csrc/fused_layout.h"
#include "../../rtl-power-fftw_amd/csrc/mixed_core.h"

namespace {

using rpf::cf;

template <class G, int J>
void load_tw(int t, const std::vector<cf>& twN, cf (*tw)[G::P - 1])
{
    if constexpr (J < G::NPASS) {
        for (int r = 1; r < G::P; ++r) tw[J - 1][r - 1] = twN[rpf::twiddle_index<G, J>(t, r)];
        load_tw<G, J + 1>(t, twN, tw);
    }
}

// run passes J..NPASS-1 for all threads of one frame, phase by phase
template <class G, int J>
void middle(std::vector<std::vector<cf>>& regs, std::vector<cf>& slab,
            const std::vector<std::vector<cf>>& tws)
{
    if constexpr (J < G::NPASS) {
        constexpr int P = G::P, T = G::T;
        for (int t = 0; t < T; ++t) {
            if constexpr (J > 1) rpf::phase_fetch<G, J>(t, regs[t].data(), slab.data());
        }
        for (int t = 0; t < T; ++t) {
            rpf::phase_butterfly_twiddle<G>(regs[t].data(), tws[t].data() + (J - 1) * (P - 1));
            rpf::phase_store<G, J>(t, regs[t].data(), slab.data());
        }
        middle<G, J + 1>(regs, slab, tws);
    }
}

template <int N, int P>
int run(const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    using G = rpf::Geom<N, P>;
    constexpr int T = G::T;
    std::vector<cf> twN(N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N; ++k) {
        long double a = two_pi * k / N;
        twN[k] = {(float)cosl(a), (float)(-sinl(a))};
    }
    std::vector<std::vector<cf>> tws(T, std::vector<cf>((G::NPASS - 1) * (P - 1)));
    for (int t = 0; t < T; ++t)
        load_tw<G, 1>(t, twN, reinterpret_cast<cf(*)[P - 1]>(tws[t].data()));
    std::vector<std::vector<cf>> regs(T, std::vector<cf>(P));
    std::vector<std::vector<double>> acc(T, std::vector<double>(P, 0.0));
    std::vector<cf> slab(G::LDS_CPX);
    // poison the padding slots: nothing may ever read them
    for (auto& c : slab) c = {NAN, NAN};

    // One emulated "workgroup" = max(T, 64) threads: whole wavefronts stage the raw
    // bytes (raw_source), exactly as the kernel does; frames beyond the first
    // slot of a wave (T < 64) are staged too but only slot 0 is computed here.
    constexpr int WAVES = T >= 64 ? T / 64 : 1;
    std::vector<uint8_t> raw(WAVES * rpf::kRawChunk * P);
    for (long f = 0; f < nframes; ++f) {
        const uint8_t* frame = stream + (size_t)f * 2 * N;
        for (int w = 0; w < WAVES; ++w)
            for (int j = 0; j < rpf::kRawChunk * P; ++j) {
                int slot, off;
                rpf::raw_source<G>(w, j, &slot, &off);
                raw[w * rpf::kRawChunk * P + j] = slot == 0 ? frame[off] : 0;
            }
        for (int t = 0; t < T; ++t) {
            const float sgn = (t & 1) ? -1.0f : 1.0f;
            float wsgn[P];
            if (window)
                for (int a = 0; a < P; ++a) wsgn[a] = window[t + T * a] * sgn;
            const uint8_t* lane_raw = raw.data() + (t / 64) * rpf::kRawChunk * P + 2 * (t % 64);
            if (window) rpf::phase_unpack<G, true>(lane_raw, sgn, wsgn, regs[t].data());
            else rpf::phase_unpack<G, false>(lane_raw, sgn, wsgn, regs[t].data());
        }
        middle<G, 1>(regs, slab, tws);
        for (int t = 0; t < T; ++t) {
            rpf::phase_fetch<G, G::NPASS>(t, regs[t].data(), slab.data());
            rpf::phase_last<G>(regs[t].data());
            rpf::phase_accumulate(regs[t].data(), acc[t].data(), P);
        }
    }
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < P; ++a) pwr[rpf::bin_of<G>(t, a)] = acc[t][a];
    return 0;
}

// Bluestein kernel (bluestein_kernel in rpf_kernels.hip), same phase order.
template <int M, int P>
int run_bluestein(int N, const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    using G = rpf::Geom<M, P>;
    constexpr int T = G::T;
    std::vector<cf> twM(M);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < M; ++k) {
        long double a = two_pi * k / M;
        twM[k] = {(float)cosl(a), (float)(-sinl(a))};
    }
    std::vector<float> g, bhat;
    rpf::make_bluestein_tables(N, window, g, bhat);
    std::vector<std::vector<cf>> tws(T, std::vector<cf>((G::NPASS - 1) * (P - 1)));
    for (int t = 0; t < T; ++t)
        load_tw<G, 1>(t, twM, reinterpret_cast<cf(*)[P - 1]>(tws[t].data()));
    std::vector<std::vector<cf>> regs(T, std::vector<cf>(P));
    std::vector<std::vector<double>> acc(T, std::vector<double>(P, 0.0));
    std::vector<cf> slab(G::LDS_CPX);
    for (long f = 0; f < nframes; ++f) {
        const uint8_t* frame = stream + (size_t)f * 2 * N;
        for (int t = 0; t < T; ++t)
            for (int a = 0; a < P; ++a) {
                const int n = t + T * a;
                regs[t][a] = cf{0.0f, 0.0f};
                if (n < N) {
                    const cf v = cf{(float)frame[2 * n] - 127.0f, (float)frame[2 * n + 1] - 127.0f};
                    regs[t][a] = rpf::cmul(v, cf{g[2 * n], g[2 * n + 1]});
                }
            }
        middle<G, 1>(regs, slab, tws);
        for (int t = 0; t < T; ++t) {
            rpf::phase_fetch<G, G::NPASS>(t, regs[t].data(), slab.data());
            rpf::phase_last<G>(regs[t].data());
        }
        for (int t = 0; t < T; ++t)
            for (int a = 0; a < P; ++a) {
                const int j = rpf::bin_of<G>(t, a);
                cf z = rpf::cmul(regs[t][a], cf{bhat[2 * j], bhat[2 * j + 1]});
                z.y = -z.y;
                slab[G::slot(j)] = z;
            }
        for (int t = 0; t < T; ++t) rpf::phase_fetch<G, 1>(t, regs[t].data(), slab.data());
        middle<G, 1>(regs, slab, tws);
        for (int t = 0; t < T; ++t) {
            rpf::phase_fetch<G, G::NPASS>(t, regs[t].data(), slab.data());
            rpf::phase_last<G>(regs[t].data());
            rpf::phase_accumulate(regs[t].data(), acc[t].data(), P);
        }
    }
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < P; ++a) {
            const int bin = rpf::bin_of<G>(t, a);
            if (bin < N) pwr[bin] = acc[t][a];
        }
    return 0;
}

// KM (mixed_plan_kernel in rpf_mixed.hip): the same per-thread functions (mixed_core.h), one
// "thread" after the other, a barrier between passes.
template <class PL, int I>
void mixed_passes(std::vector<std::vector<cf>>& tws, std::vector<cf>& slab, const uint8_t* frame,
                  const float* window, std::vector<std::vector<double>>& acc)
{
    if constexpr (I < PL::F) {
        constexpr int R = PL::R(I), G = PL::G(I);
        for (int t = 0; t < PL::TPF(I); ++t)
            for (int g = 0; g < G; ++g) {
                cf v[R];
                if constexpr (I == 0) {
                    uint32_t raw[(R + 1) / 2] = {};
                    float wsgn[R];
                    for (int n1 = 0; n1 < R; ++n1) {
                        const int n = rpf::mix_sample_index<PL>(t, g, n1);
                        raw[n1 >> 1] |= (frame[2 * n] | (uint32_t)frame[2 * n + 1] << 8) << (16 * (n1 & 1));
                        wsgn[n1] = window ? window[n] * ((n & 1) ? -1.0f : 1.0f) : 0.0f;
                    }
                    const float sgn = ((t + g * PL::TPF(0)) & 1) ? -1.0f : 1.0f;
                    if (window) rpf::mix_unpack<PL, true, 0>(raw, sgn, wsgn, v);
                    else rpf::mix_unpack<PL, false, 0>(raw, sgn, wsgn, v);
                } else {
                    rpf::mix_fetch<PL, I>(rpf::mix_slot_base<PL, I>(t, g), v, slab.data());
                }
                if constexpr (I < PL::F - 1) {
                    rpf::mix_butterfly<PL, I>(v, tws[t].data() + PL::tw_offset(I) + g * (R - 1));
                    rpf::mix_store<PL, I>(rpf::mix_slot_base<PL, I>(t, g), v, slab.data());
                } else {
                    rpf::mix_last_pass_accumulate<PL>(v, acc[t].data() + g * R);       // (float, or double for the wide plans)
                }
            }
        mixed_passes<PL, I + 1>(tws, slab, frame, window, acc);
    }
}
template <class PL, int I>
void mixed_load_tw(int t, const std::vector<cf>& twN, cf* tw)
{
    if constexpr (I < PL::F - 1) {
        if (t < PL::TPF(I))
            for (int g = 0; g < PL::G(I); ++g)
                for (int k = 1; k < PL::R(I); ++k)
                    tw[PL::tw_offset(I) + g * (PL::R(I) - 1) + k - 1] = twN[rpf::mix_twiddle_index<PL, I>(t, g, k)];
        mixed_load_tw<PL, I + 1>(t, twN, tw);
    }
}
template <class PL>
int run_mixed(const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    constexpr int N = PL::N;
    std::vector<cf> twN(N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N; ++k) {
        long double a = two_pi * k / N;
        twN[k] = {(float)cosl(a), (float)(-sinl(a))};
    }
    std::vector<std::vector<cf>> tws(PL::TPFMAX, std::vector<cf>(PL::NTW + 1));
    for (int t = 0; t < PL::TPFMAX; ++t) mixed_load_tw<PL, 0>(t, twN, tws[t].data());
    std::vector<std::vector<double>> acc(PL::TPFMAX, std::vector<double>(PL::PPTL, 0.0));
    std::vector<cf> slab(PL::LDS_CPX);
    for (long f = 0; f < nframes; ++f)
        mixed_passes<PL, 0>(tws, slab, stream + (size_t)f * 2 * N, window, acc);
    std::vector<int> seen(N, 0);
    for (int t = 0; t < PL::TPF(PL::F - 1); ++t)
        for (int g = 0; g < PL::G(PL::F - 1); ++g)
            for (int k = 0; k < PL::RLAST; ++k) {
                const int bin = rpf::mix_bin<PL>(t, g, k);
                if (bin < 0 || bin >= N || seen[bin]++) return -2;
                pwr[bin] = acc[t][g * PL::RLAST + k];
            }
    return 0;
}

// one pair of sections of the paired split form, sample by sample (the kernel: split_pair_rolling)
template <class PL, bool FIRST, int I = 0>
void pair_section(bool windowed, const uint32_t* ra, const uint32_t* rb, const float* sgn, float sp, const float* wa,
                  const float* wb, cf wpj, cf* v)
{
    if constexpr (I < PL::PPT0) {
        if (windowed) rpf::mix_split_pair_element<PL, 2, FIRST, I>(ra[I], rb[I], sgn, sp, wa[I], wb[I], wpj, v);
        else rpf::mix_split_pair_element<PL, 0, FIRST, I>(ra[I], rb[I], sgn, sp, 0.0f, 0.0f, wpj, v);
        pair_section<PL, FIRST, I + 1>(windowed, ra, rb, sgn, sp, wa, wb, wpj, v);
    }
}

// The split form (N = P M, workgroup p computes X[p + P k] with the M-point plan): pass 0 through
// mix_unpack_split / mix_butterfly_split, the later passes as above.
template <class PL, int P>
int run_mixed_split(const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    constexpr int M = PL::N, N = P * M, R0 = PL::R(0), G0 = PL::G(0);
    std::vector<cf> twN(N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N; ++k) {
        long double a = two_pi * k / N;
        twN[k] = {(float)cosl(a), (float)(-sinl(a))};
    }
    std::vector<cf> twM(M);
    for (int k = 0; k < M; ++k) twM[k] = twN[(size_t)P * k];
    for (int p = 0; p < P; ++p) {
        cf wp[P], mid[R0];
        for (int j = 0; j < P; ++j) wp[j] = twN[((long)j * p * M) % N];
        for (int n1 = 0; n1 < R0; ++n1) mid[n1] = twN[((long)n1 * PL::S(0) * p) % N];
        std::vector<std::vector<cf>> tws(PL::TPFMAX, std::vector<cf>(PL::NTW + 1));
        for (int t = 0; t < PL::TPFMAX; ++t) mixed_load_tw<PL, 0>(t, twM, tws[t].data());
        std::vector<std::vector<double>> acc(PL::TPFMAX, std::vector<double>(PL::PPTL, 0.0));
        std::vector<cf> slab(PL::LDS_CPX);
        for (long f = 0; f < nframes; ++f) {
            const uint8_t* frame = stream + (size_t)f * 2 * N;
            for (int t = 0; t < PL::TPF(0); ++t) {
                cf v[PL::PPT0];
                float sgn[G0];
                for (int g = 0; g < G0; ++g) sgn[g] = ((t + g * PL::TPF(0)) & 1) ? -1.0f : 1.0f;
                auto load = [&](int j, uint32_t* raw) {
                    for (int i = 0; i < PL::PPT0; ++i) {
                        const int n = rpf::mix_sample_index<PL>(t, i / R0, i % R0) + j * M;
                        raw[i] = frame[2 * n] | (uint32_t)frame[2 * n + 1] << 8;
                    }
                };
                if constexpr (P > 5) {          // the paired form: sections j and j + P/2 first (mix_split_pair_element)
                    constexpr int Q = P / 2;
                    const float sp = (p & 1) ? -1.0f : 1.0f;
                    for (int j = 0; j < Q; ++j) {
                        uint32_t ra[PL::PPT0], rb[PL::PPT0];
                        float wa[PL::PPT0], wb[PL::PPT0];
                        load(j, ra);
                        load(j + Q, rb);
                        for (int i = 0; i < PL::PPT0; ++i) {
                            const int n = rpf::mix_sample_index<PL>(t, i / R0, i % R0);
                            wa[i] = window ? window[n + j * M] : 0.0f;
                            wb[i] = window ? window[n + (j + Q) * M] : 0.0f;
                        }
                        if (j == 0) pair_section<PL, true>(window != nullptr, ra, rb, sgn, sp, wa, wb, wp[j], v);
                        else pair_section<PL, false>(window != nullptr, ra, rb, sgn, sp, wa, wb, wp[j], v);
                    }
                } else
                for (int j = 0; j < P; ++j) {
                    uint32_t raw[PL::PPT0];
                    load(j, raw);
                    const float* w = window ? window + j * M + t : nullptr;
                    if (j == 0) {
                        if (window) rpf::mix_split_accumulate<PL, true, true>(raw, sgn, w, wp[j], v);
                        else rpf::mix_split_accumulate<PL, false, true>(raw, sgn, w, wp[j], v);
                    } else {
                        if (window) rpf::mix_split_accumulate<PL, true, false>(raw, sgn, w, wp[j], v);
                        else rpf::mix_split_accumulate<PL, false, false>(raw, sgn, w, wp[j], v);
                    }
                }
                rpf::mix_split_mid<PL>(v, mid);
                for (int g = 0; g < G0; ++g) {
                    const int ntail = t + g * PL::TPF(0);
                    cf tw0[R0];
                    for (int k = 0; k < R0; ++k) tw0[k] = twN[((long)ntail * (p + P * k)) % N];
                    rpf::mix_butterfly_split<PL>(v + g * R0, tw0);
                    rpf::mix_store<PL, 0>(rpf::mix_slot_base<PL, 0>(t, g), v + g * R0, slab.data());
                }
            }
            mixed_passes<PL, 1>(tws, slab, frame, window, acc);
        }
        for (int t = 0; t < PL::TPF(PL::F - 1); ++t)
            for (int g = 0; g < PL::G(PL::F - 1); ++g)
                for (int k = 0; k < PL::RLAST; ++k) pwr[p + P * rpf::mix_bin<PL>(t, g, k)] = acc[t][g * PL::RLAST + k];
    }
    return 0;
}

}  // namespace

using rpf::MixPlan;
using rpf::MPass;
extern "C" int rpf_emul_mixed(int plan, const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    switch (plan) {
        case 0: return run_mixed<MixPlan<100, 1, 0, MPass<10>, MPass<10>>>(window, stream, nframes, pwr);
        case 1: return run_mixed<MixPlan<500, 1, 0, MPass<10>, MPass<10>, MPass<5, 2>>>(window, stream, nframes, pwr);
        case 2: return run_mixed<MixPlan<500, 1, 0, MPass<25>, MPass<20>>>(window, stream, nframes, pwr);
        case 3: return run_mixed<MixPlan<1000, 1, 0, MPass<10>, MPass<10>, MPass<10>>>(window, stream, nframes, pwr);
        case 4: return run_mixed<MixPlan<1200, 1, 0, MPass<10>, MPass<12>, MPass<10>>>(window, stream, nframes, pwr);
        case 5: return run_mixed<MixPlan<300, 1, 0, MPass<5, 2>, MPass<6, 2>, MPass<10>>>(window, stream, nframes, pwr);
        case 6: return run_mixed<MixPlan<3600, 1, 0, MPass<15>, MPass<16>, MPass<15>>>(window, stream, nframes, pwr);
        case 7: return run_mixed<MixPlan<1080, 1, 0, MPass<9>, MPass<8>, MPass<15>>>(window, stream, nframes, pwr);
        case 8: return run_mixed<MixPlan<6000, 1, 0, MPass<20>, MPass<20>, MPass<15>>>(window, stream, nframes, pwr);
        case 9: return run_mixed<MixPlan<96, 1, 0, MPass<2, 3>, MPass<3, 2>, MPass<4>, MPass<4>>>(window, stream, nframes, pwr);
        case 10: return run_mixed<MixPlan<700, 1, 0, MPass<7, 2>, MPass<10>, MPass<10>>>(window, stream, nframes, pwr);
        case 11: return run_mixed<MixPlan<2860, 1, 0, MPass<13>, MPass<11>, MPass<20>>>(window, stream, nframes, pwr);
        case 12: return run_mixed<MixPlan<782, 1, 0, MPass<17>, MPass<23>, MPass<2, 17>>>(window, stream, nframes, pwr);
        case 13: return run_mixed_split<MixPlan<500, 1, 0, MPass<10>, MPass<10>, MPass<5, 2>>, 2>(window, stream, nframes, pwr);
        case 14: return run_mixed_split<MixPlan<1000, 1, 0, MPass<10>, MPass<10>, MPass<10>>, 5>(window, stream, nframes, pwr);
        case 15: return run_mixed_split<MixPlan<600, 1, 0, MPass<25>, MPass<24>>, 3>(window, stream, nframes, pwr);
        case 16: return run_mixed_split<MixPlan<96, 1, 0, MPass<2, 3>, MPass<3, 2>, MPass<4>, MPass<4>>, 4>(window, stream, nframes, pwr);
        // the paired form (P = 6, 8, 10)
        case 17: return run_mixed_split<MixPlan<100, 1, 0, MPass<10>, MPass<10>>, 6>(window, stream, nframes, pwr);
        case 18: return run_mixed_split<MixPlan<96, 1, 0, MPass<2, 3>, MPass<3, 2>, MPass<4>, MPass<4>>, 8>(window, stream, nframes, pwr);
        case 19: return run_mixed_split<MixPlan<500, 1, 0, MPass<10>, MPass<10>, MPass<5, 2>>, 10>(window, stream, nframes, pwr);
    }
    return -1;
}
// every plan the product ships (the kernel tables themselves, with the emulator's runners in place of the kernels)
struct ShippedPlan {
    int N;
    int (*run)(const float*, const uint8_t*, long, double*);
};
template <class PL>
constexpr ShippedPlan plan_entry(int)
{
    return {PL::N, &run_mixed<PL>};
}
// (window mode and pipeline shape of the kernel's windowed twin: the arithmetic is the same)
template <int SPLIT, class PL, int WM = 0, bool ROLL = false>
constexpr ShippedPlan split_entry(int)
{
    return {SPLIT * PL::N, &run_mixed_split<rpf::SplitPlan<SPLIT, PL>, SPLIT>};      // (float or wide last pass: mixed_core.h)
}
template <int R, int G = 1>
using P = MPass<R, G>;
const ShippedPlan kShipped[] = {
#include "../../rtl-power-fftw_amd/csrc/mixed_plans.inc"
#include "../../rtl-power-fftw_amd/csrc/mixed_plans_split.inc"
};
extern "C" int rpf_emul_shipped_count() { return (int)(sizeof kShipped / sizeof kShipped[0]); }
extern "C" int rpf_emul_shipped_n(int i) { return kShipped[i].N; }
extern "C" int rpf_emul_shipped(int i, const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    return kShipped[i].run(window, stream, nframes, pwr);
}

extern "C" int rpf_emul_mixed_n(int plan)
{
    const int n[] = {100, 500, 500, 1000, 1200, 300, 3600, 1080, 6000, 96, 700, 2860, 782, 1000, 5000, 1800, 384, 600, 768, 5000};
    return plan >= 0 && plan < 20 ? n[plan] : -1;
}

// v[k] <- sum_n v[n] W_R^{nk} through dft_small.h (interleaved re, im)
extern "C" int rpf_emul_small_dft(int R, float* v)
{
    cf* c = reinterpret_cast<cf*>(v);
    switch (R) {
#define CASE(r) case r: rpf::SmallDft<r>::run(c); return 0
        CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7); CASE(8); CASE(9); CASE(10); CASE(11); CASE(12); CASE(13); CASE(14);
        CASE(15); CASE(16); CASE(17); CASE(18); CASE(19); CASE(20); CASE(21); CASE(22); CASE(23); CASE(24); CASE(25);
#undef CASE
    }
    return -1;
}

// the same through dft_small_wide.h (double; interleaved re, im): the split forms' wide last passes
extern "C" int rpf_emul_wide_dft(int R, double* v)
{
    rpf::cd* c = reinterpret_cast<rpf::cd*>(v);
    switch (R) {
#define CASE(r) case r: rpf::WideDft<r>::run(c); return 0
        CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7); CASE(8); CASE(9); CASE(10); CASE(11); CASE(12); CASE(13); CASE(14);
        CASE(15); CASE(16); CASE(17); CASE(18); CASE(19); CASE(20); CASE(21); CASE(22); CASE(23); CASE(24); CASE(25);
#undef CASE
    }
    return -1;
}

// the fused wide pass (WidePass: float values in, every output handed to a functor): out[k] = X_k, interleaved re, im
struct CollectWide {
    double* out;
    void operator()(int k, rpf::cd x) const { out[2 * k] = x.x; out[2 * k + 1] = x.y; }
};
extern "C" int rpf_emul_wide_pass(int R, const float* v, double* out)
{
    const cf* c = reinterpret_cast<const cf*>(v);
    switch (R) {
#define CASE(r) case r: rpf::WidePass<r>::run(c, CollectWide{out}); return 0
        CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7); CASE(8); CASE(9); CASE(10); CASE(11); CASE(12); CASE(13); CASE(14);
        CASE(15); CASE(16); CASE(17); CASE(18); CASE(19); CASE(20); CASE(21); CASE(22); CASE(23); CASE(24); CASE(25);
#undef CASE
    }
    return -1;
}

extern "C" int rpf_emul_accumulate(int N, int P, const float* window, const uint8_t* stream,
                                   long nframes, double* pwr)
{
#define CASE(n, p) if (N == n && P == p) return run<n, p>(window, stream, nframes, pwr)
    CASE(64, 8); CASE(128, 8); CASE(256, 8); CASE(512, 8); CASE(1024, 8); CASE(4096, 8);
    CASE(1024, 16); CASE(2048, 16); CASE(4096, 16); CASE(8192, 16); CASE(256, 16); CASE(512, 16); CASE(128, 16);
#undef CASE
    return -1;
}

extern "C" int rpf_emul_bluestein(int N, const float* window, const uint8_t* stream, long nframes, double* pwr)
{
    switch (rpf::bluestein_length(N)) {
        case 64: return run_bluestein<64, 8>(N, window, stream, nframes, pwr);
        case 128: return run_bluestein<128, 8>(N, window, stream, nframes, pwr);
        case 256: return run_bluestein<256, 8>(N, window, stream, nframes, pwr);
        case 512: return run_bluestein<512, 8>(N, window, stream, nframes, pwr);
        case 1024: return run_bluestein<1024, 8>(N, window, stream, nframes, pwr);
        case 2048: return run_bluestein<2048, 16>(N, window, stream, nframes, pwr);
        case 4096: return run_bluestein<4096, 16>(N, window, stream, nframes, pwr);
        case 8192: return run_bluestein<8192, 16>(N, window, stream, nframes, pwr);
    }
    return -1;
}

// ---- hop_partition.h: the host-side partition of a scan's hops over the workgroups of one
// launch, and the kernel's walk over it restated (same HopCursor, same loop skeleton as
// fft_accum_kernel) so that the CPU tests can check that every frame of every hop is visited
// exactly once and that every partial slot is written exactly once.
#include "../../rtl-power-fftw_amd/csrc/hop_partition.h"

// out_visits[h][f] += 1 for every (hop, frame) a workgroup accumulates; out_slot_hop[slot] = the hop
// whose partial lands in `slot` (-1: never written; -2: written twice); out_staged[h][f] += 1 for
// every (hop, frame) staged by slot-0 frame index (what the DMA ring fetched, clamped frames included).
// Returns the grid, or a negative number on inconsistency.
extern "C" int rpf_emul_walk_hops(const int64_t* nframes, int H, int fpw, int max_grid, int rawd, int interleave_single,
                                  int* it_begin, int* slot_begin, int32_t* const* out_visits,
                                  int* out_slot_hop, int nslots_cap, long* staged_iterations)
{
    rpf::HopArgs a;
    rpf::SlotRanges r;
    const int grid = rpf::partition_hops(nframes, H, fpw, max_grid, &a, &r, interleave_single != 0);
    if (grid < 0) return grid;
    for (int h = 0; h <= H; ++h) {
        it_begin[h] = a.it_begin[h];
        slot_begin[h] = r.begin[h];
    }
    if (r.begin[H] > nslots_cap) return -10;
    for (int s = 0; s < nslots_cap; ++s) out_slot_hop[s] = -1;
    for (int h = 0; h < rpf::kMaxHops; ++h) a.stream[h] = nullptr;
    *staged_iterations = 0;
    const rpf::HopArgsView tbl{a};
    if (a.total != a.it_begin[a.H] || (grid && a.q * grid + a.r != a.total)) return -14;
    if (grid && a.step != 1 && a.step != grid) return -16;
    long iterations = 0;
    for (int w = 0; w < grid; ++w) {
        int first, count;
        rpf::hop_share(w, a.q, a.r, a.step, &first, &count);
        if (count < 1) return -11;                       // the host promises every workgroup an iteration
        iterations += count;
        // (the kernel's control flow, statement for statement: rpf_kernels.hip, fft_accum_kernel)
        const int step = a.step, fstep = fpw * step;
        rpf::HopCursor ahead;
        ahead.seek(tbl, first);
        int ahead_fb = (ahead.j - ahead.begin) * fpw, ahead_fstep = fstep;
        int ahead_left = count;
        auto run_length = [&](const rpf::HopCursor& c, int left) {
            const int in_hop = step == 1 ? c.end - c.j : left;
            return in_hop < left ? in_hop : left;
        };
        int ahead_run = run_length(ahead, ahead_left);
        ahead_left -= ahead_run;
        bool parked = false;
        auto ahead_turn = [&]() {
            if (ahead_left > 0) {
                ahead.seek(tbl, ahead.end);
                ahead_fb = 0;
                ahead_run = run_length(ahead, ahead_left);
                ahead_left -= ahead_run;
            } else {
                ahead.seek(tbl, 0);
                ahead_fb = 0;
                ahead_fstep = 0;
                ahead_run = 0x7fffffff;
                parked = true;
            }
        };
        std::vector<std::pair<int, int>> ring(rawd);     // what each ring slot holds: (hop, slot-0 frame); parked: (-1, 0)
        auto stage_next = [&](int slot) {
            ring[slot] = parked ? std::make_pair(-1, 0) : std::make_pair(ahead.h, ahead_fb);
            ++*staged_iterations;
            ahead_fb += ahead_fstep;
            if (--ahead_run == 0) ahead_turn();
        };
        for (int d = 0; d < rawd; ++d) stage_next(d);
        rpf::HopCursor cur;
        cur.seek(tbl, first);
        int it = 0;
        while (true) {
            const int seg = run_length(cur, count - it);
            if (seg < 1) return -17;
            int fb = (cur.j - cur.begin) * fpw;
            for (int n = seg; n > 0; --n, ++it, fb += fstep) {
                if (ring[it % rawd] != std::make_pair(cur.h, fb)) return -12;   // the ring holds another iteration's bytes
                if (fb >= cur.nframes) return -18;                             // an iteration without a frame
                for (int fs = 0; fs < fpw; ++fs)
                    if (fb + fs < cur.nframes) out_visits[cur.h][fb + fs] += 1;
                stage_next(it % rawd);
            }
            const int slot = a.slot_bias[cur.h] + w;
            if (slot < r.begin[cur.h] || slot >= r.begin[cur.h + 1]) return -13;
            out_slot_hop[slot] = out_slot_hop[slot] == -1 ? cur.h : -2;
            if (it >= count) break;
            cur.seek(tbl, cur.end);
        }
    }
    if (iterations != a.total) return -15;
    return grid;
}

// The fused four-step kernel (rpf_fourstep.hip) factors the inter-step twiddle W_N^{c bin_of(t, a)} into a per-lane
// register W_N^{c bin_of(t, 0)} and a per-register LDS value W_N^{c bin_of(0, a)}: that needs
// bin_of(t, a) == bin_of(t, 0) + bin_of(0, a) for every lane and register of the column geometries it runs
// (Geom<128 | 256 | 512, 8>: the digits of 8 t + a do not carry).  Returns the number of (t, a) that break it.
template <class G>
static int bin_split_violations()
{
    int bad = 0;
    for (int t = 0; t < G::T; ++t)
        for (int a = 0; a < G::P; ++a)
            if (rpf::bin_of<G>(t, a) != rpf::bin_of<G>(t, 0) + rpf::bin_of<G>(0, a)) ++bad;
    return bad;
}
extern "C" int rpf_emul_fused_bin_split(int n1)
{
    switch (n1) {
        case 128: return bin_split_violations<rpf::Geom<128, 8>>();
        case 256: return bin_split_violations<rpf::Geom<256, 8>>();
        case 512: return bin_split_violations<rpf::Geom<512, 8>>();
        default: return -1;
    }
}

// The fused four-step kernel's raw-row staging (fused_layout.h): every LDS piece q of a tile of `rows` rows of `rowb`
// bytes holds a (row, piece) nobody else holds, and the reader's offset of every byte of that piece is where the
// writer's lane put it (16 q + byte % 16).  Returns the number of violations, -1 for a row length there is no kernel for.
template <int ROWB>
static int raw_stage_violations(int rows)
{
    using R = rpf::RawStage<ROWB>;
    const int pieces = rows * R::PPR;
    if (pieces % 64 != 0) return -1;
    std::vector<int> seen(pieces, 0);
    int bad = 0;
    for (int q = 0; q < pieces; ++q) {
        const int row = R::row_of(q), piece = R::piece_of(q);
        if (row < 0 || row >= rows || piece < 0 || piece >= R::PPR) { ++bad; continue; }
        ++seen[row * R::PPR + piece];
        for (int b = 0; b < 16; ++b)
            if (R::offset(row, 16 * piece + b) != 16 * q + b) ++bad;
    }
    for (int v : seen) bad += v != 1;
    return bad;
}
extern "C" int rpf_emul_fused_raw_stage(int rowb, int rows)
{
    switch (rowb) {
        case 32: return raw_stage_violations<32>(rows);
        case 64: return raw_stage_violations<64>(rows);
        case 128: return raw_stage_violations<128>(rows);
        default: return -1;
    }
}
// bank (of 32, ds_read_b32) of the dword the lane of row `row` reads at byte `byte` of its row
extern "C" int rpf_emul_fused_raw_bank(int rowb, int row, int byte)
{
    switch (rowb) {
        case 32: return (rpf::RawStage<32>::offset(row, byte) / 4) % 32;
        case 64: return (rpf::RawStage<64>::offset(row, byte) / 4) % 32;
        case 128: return (rpf::RawStage<128>::offset(row, byte) / 4) % 32;
        default: return -1;
    }
}
