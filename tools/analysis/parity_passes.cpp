// parity_passes.cpp -- where does the split / paired mixed-radix form lose accuracy against the CPU path?
//   CPU only.  Runs a shipped split-form plan thread by thread through the kernel's own
// per-thread functions (mixed_core.h, dft_small.h -- like tests/emul), with any of its stages replaced by an
// "ideal" twin computed in long double and rounded ONCE per element:
//   stage 0       section sums (first radix-P pass for one residue) + the mid twiddle W_N^{n1 S_0 p}
//   stage 1       pass 0 of the M-point plan: DFT_{R_0} and the folded twiddles W_N^{ntail (p + P k)}
//   stage 1 + i   pass i: DFT_{R_i} and the table twiddles W_M^{D_i ntail k} (none in the last pass)
// mode per stage: 0 = as shipped; 1 = exact arithmetic on the FLOAT twiddle values the kernel uses (what is left is the
// twiddles' representation error); 2 = exact arithmetic, exact twiddles; 3 = as shipped but every twiddle product
// of the stage compensated (twiddle = float hi + float lo, two more FMAs per component); 4 = exact arithmetic, exact
// inter-pass twiddles, but the DFT matrix entries W_R^{nk} rounded to float (the butterfly's constants' representation
// error alone); 5 (last stage) = the wide last pass the kernels can run (dft_small_wide.h: double butterfly, squares in double)
// Built and driven by tools/analysis/parity_passes.py.  Not a product path.
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

#include "../../rtl-power-fftw_amd/csrc/fft_core.h"
#include "../../rtl-power-fftw_amd/csrc/mixed_core.h"

namespace {
using rpf::cf;
using ld = long double;
using cld = std::complex<long double>;
const ld kTwoPi = 6.283185307179586476925286766559005768L;

cld wexact(long num, long den)
{
    num %= den;
    const ld a = kTwoPi * static_cast<ld>(num) / static_cast<ld>(den);
    return cld(cosl(a), -sinl(a));
}
cf to_cf(cld z) { return cf{static_cast<float>(z.real()), static_cast<float>(z.imag())}; }
cld to_ld(cf z) { return cld(z.x, z.y); }
cf wfloat(long num, long den) { return to_cf(wexact(num, den)); }
// the twiddle the stage multiplies with: float value (modes 0, 1), exact (2)
cld tw_for(int mode, long num, long den) { return mode == 2 ? wexact(num, den) : to_ld(wfloat(num, den)); }
// compensated product a * (hi + lo) in float arithmetic: hi = float(w), lo = float(w - hi)
cf cmul_comp(cf a, long num, long den)
{
    const cld w = wexact(num, den);
    const cf hi = to_cf(w);
    const cf lo = to_cf(w - to_ld(hi));
    const cf r = rpf::cmul(a, hi);
    return cf{__builtin_fmaf(-a.y, lo.y, __builtin_fmaf(a.x, lo.x, r.x)), __builtin_fmaf(a.x, lo.y, __builtin_fmaf(a.y, lo.x, r.y))};
}

template <class PL, int I>
void later_passes(int p, int P, const int* mode, std::vector<cf>& slab, std::vector<std::vector<double>>& acc)
{
    if constexpr (I < PL::F) {
        constexpr int R = PL::R(I), G = PL::G(I), M = PL::N;
        const int m = mode[1 + I];
        for (int t = 0; t < PL::TPF(I); ++t)
            for (int g = 0; g < G; ++g) {
                cf v[R];
                rpf::mix_fetch<PL, I>(rpf::mix_slot_base<PL, I>(t, g), v, slab.data());
                const int b = t + g * PL::TPF(I);
                const long ntail = b % PL::S(I);
                if (m == 1 || m == 2 || m == 4) {
                    cld in[R];
                    for (int n = 0; n < R; ++n) in[n] = to_ld(v[n]);
                    for (int k = 0; k < R; ++k) {
                        cld s = 0;
                        for (int n = 0; n < R; ++n)
                            s += in[n] * (m == 4 ? to_ld(wfloat(static_cast<long>(n) * k, R)) : wexact(static_cast<long>(n) * k, R));
                        if (I < PL::F - 1 && k > 0) s *= tw_for(m == 4 ? 2 : m, static_cast<long>(PL::D(I)) * ntail * k, M);
                        v[k] = to_cf(s);
                    }
                } else if (m == 5 && I == PL::F - 1) {
                    rpf::cd w[R];
                    for (int n = 0; n < R; ++n) w[n] = rpf::cd{static_cast<double>(v[n].x), static_cast<double>(v[n].y)};
                    rpf::WideDft<R>::run(w);
                    double* a = acc[t].data() + g * R;
                    for (int k = 0; k < R; ++k) a[k] = __builtin_fma(w[k].y, w[k].y, __builtin_fma(w[k].x, w[k].x, a[k]));
                    continue;
                } else if (m == 3 && I < PL::F - 1) {
                    rpf::SmallDft<R>::run(v);
                    for (int k = 1; k < R; ++k) v[k] = cmul_comp(v[k], static_cast<long>(PL::D(I)) * ntail * k, M);
                } else {
                    cf tw[R];
                    for (int k = 1; k < R; ++k) tw[k - 1] = wfloat(static_cast<long>(PL::D(I)) * ntail * k, M);
                    rpf::mix_butterfly<PL, I>(v, tw);
                }
                if constexpr (I < PL::F - 1) rpf::mix_store<PL, I>(rpf::mix_slot_base<PL, I>(t, g), v, slab.data());
                else rpf::phase_accumulate(v, acc[t].data() + g * R, R);
            }
        later_passes<PL, I + 1>(p, P, mode, slab, acc);
    }
}

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

template <class PL, int P>
int run_split(const float* window, const uint8_t* stream, long nframes, const int* mode, double* pwr)
{
    constexpr int M = PL::N, N = P * M, R0 = PL::R(0), G0 = PL::G(0);
    for (int p = 0; p < P; ++p) {
        cf wp[P], mid[R0];
        for (int j = 0; j < P; ++j) wp[j] = wfloat(static_cast<long>(j) * p * M, N);
        for (int n1 = 0; n1 < R0; ++n1) mid[n1] = wfloat(static_cast<long>(n1) * PL::S(0) * p, N);
        std::vector<std::vector<double>> acc(PL::TPFMAX, std::vector<double>(PL::PPTL, 0.0));
        std::vector<cf> slab(PL::LDS_CPX);
        for (long f = 0; f < nframes; ++f) {
            const uint8_t* frame = stream + static_cast<size_t>(f) * 2 * N;
            for (int t = 0; t < PL::TPF(0); ++t) {
                cf v[PL::PPT0];
                float sgn[G0];
                for (int g = 0; g < G0; ++g) sgn[g] = ((t + g * PL::TPF(0)) & 1) ? -1.0f : 1.0f;
                auto load = [&](int j, uint32_t* raw) {
                    for (int i = 0; i < PL::PPT0; ++i) {
                        const int n = rpf::mix_sample_index<PL>(t, i / R0, i % R0) + j * M;
                        raw[i] = frame[2 * n] | static_cast<uint32_t>(frame[2 * n + 1]) << 8;
                    }
                };
                if (mode[0] == 1 || mode[0] == 2) {
                    for (int i = 0; i < PL::PPT0; ++i) {
                        const int n1 = i % R0;
                        const int n = rpf::mix_sample_index<PL>(t, i / R0, n1);
                        cld s = 0;
                        for (int j = 0; j < P; ++j) {
                            const int nn = n + j * M;
                            const float sg = (nn & 1) ? -1.0f : 1.0f;
                            float re = (static_cast<float>(frame[2 * nn]) - 127.0f) * sg;
                            float im = (static_cast<float>(frame[2 * nn + 1]) - 127.0f) * sg;
                            if (window) {          // the reference's one rounding (datastore.cxx:76-77)
                                re *= window[nn];
                                im *= window[nn];
                            }
                            s += cld(re, im) * tw_for(mode[0], static_cast<long>(j) * p * M, N);
                        }
                        s *= tw_for(mode[0], static_cast<long>(n1) * PL::S(0) * p, N);
                        v[i] = to_cf(s);
                    }
                } else {
                    if constexpr (P > 5) {
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
                    } else {
                        for (int j = 0; j < P; ++j) {
                            uint32_t raw[PL::PPT0];
                            load(j, raw);
                            const float* w = window ? window + j * M + t : nullptr;
                            if (j == 0) {
                                if (window) rpf::mix_split_accumulate<PL, 1, true>(raw, sgn, w, wp[j], v);
                                else rpf::mix_split_accumulate<PL, 0, true>(raw, sgn, w, wp[j], v);
                            } else {
                                if (window) rpf::mix_split_accumulate<PL, 1, false>(raw, sgn, w, wp[j], v);
                                else rpf::mix_split_accumulate<PL, 0, false>(raw, sgn, w, wp[j], v);
                            }
                        }
                    }
                    if (mode[0] == 3) {
                        for (int i = 0; i < PL::PPT0; ++i) v[i] = cmul_comp(v[i], static_cast<long>(i % R0) * PL::S(0) * p, N);
                    } else {
                        rpf::mix_split_mid<PL>(v, mid);
                    }
                }
                for (int g = 0; g < G0; ++g) {
                    const long ntail = t + g * PL::TPF(0);
                    cf* const vg = v + g * R0;
                    if (mode[1] == 1 || mode[1] == 2) {
                        cld in[R0];
                        for (int n = 0; n < R0; ++n) in[n] = to_ld(vg[n]);
                        for (int k = 0; k < R0; ++k) {
                            cld s = 0;
                            for (int n = 0; n < R0; ++n) s += in[n] * wexact(static_cast<long>(n) * k, R0);
                            s *= tw_for(mode[1], ntail * (p + static_cast<long>(P) * k), N);
                            vg[k] = to_cf(s);
                        }
                    } else if (mode[1] == 3) {
                        rpf::SmallDft<R0>::run(vg);
                        for (int k = 0; k < R0; ++k) vg[k] = cmul_comp(vg[k], ntail * (p + static_cast<long>(P) * k), N);
                    } else {
                        cf tw0[R0];
                        for (int k = 0; k < R0; ++k) tw0[k] = wfloat(ntail * (p + static_cast<long>(P) * k), N);
                        rpf::mix_butterfly_split<PL>(vg, tw0);
                    }
                    rpf::mix_store<PL, 0>(rpf::mix_slot_base<PL, 0>(t, g), vg, slab.data());
                }
            }
            later_passes<PL, 1>(p, P, mode, slab, acc);
        }
        for (int t = 0; t < PL::TPF(PL::F - 1); ++t)
            for (int g = 0; g < PL::G(PL::F - 1); ++g)
                for (int k = 0; k < PL::RLAST; ++k) pwr[p + P * rpf::mix_bin<PL>(t, g, k)] = acc[t][g * PL::RLAST + k];
    }
    return PL::F;
}

using rpf::MixPlan;
template <int R, int G = 1>
using P = rpf::MPass<R, G>;
}  // namespace

// case: the shipped form of a size (plain or windowed twin); returns the number of passes of the M-point plan, < 0: unknown
extern "C" int rpf_analysis_split(int N, int windowed_twin, const float* window, const uint8_t* stream, long nframes,
                                  const int* mode, double* pwr)
{
#define CASE(n, wt, Pn, ...) if (N == n && windowed_twin == wt) return run_split<MixPlan<__VA_ARGS__>, Pn>(window, stream, nframes, mode, pwr)
    CASE(54000, 0, 5, 10800, 1, 2, P<3, 5>, P<15>, P<16>, P<15>);
    CASE(54000, 1, 6, 9000, 1, 2, P<9>, P<10>, P<10>, P<10>);
    CASE(66000, 0, 5, 13200, 1, 2, P<22>, P<24>, P<25>);
    CASE(66000, 1, 10, 6600, 1, 2, P<10>, P<6, 2>, P<11>, P<10>);
    CASE(88000, 0, 10, 8800, 1, 2, P<20>, P<20>, P<22>);
    CASE(104000, 0, 8, 13000, 1, 2, P<5, 4>, P<13, 2>, P<20>, P<10, 2>);
    CASE(104000, 1, 10, 10400, 1, 2, P<4, 5>, P<10, 2>, P<13, 2>, P<20>);
    CASE(70000, 0, 5, 14000, 1, 2, P<10, 2>, P<7, 4>, P<10, 2>, P<20>);
    CASE(20000, 0, 2, 10000, 1, 2, P<10>, P<10>, P<10>, P<10>);
    CASE(90000, 0, 10, 9000, 1, 2, P<9>, P<10>, P<10>, P<10>);
    CASE(81000, 0, 10, 8100, 1, 2, P<3, 4>, P<15>, P<15>, P<12>);
    CASE(108000, 0, 8, 13500, 1, 2, P<15>, P<4, 5>, P<15>, P<15>);
    CASE(75000, 0, 5, 15000, 1, 2, P<10, 3>, P<10, 3>, P<10, 3>, P<15, 2>);
#undef CASE
    return -1;
}

// rms rounding error of SmallDft<R> on random inputs, relative to the rms of its outputs (and the worst single output)
namespace {
template <int R>
void radix_error(int trials, double* out)
{
    unsigned long long s = 0x9E3779B97F4A7C15ull * (R + 1);
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return static_cast<float>((s >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    };
    long double e2 = 0, o2 = 0, worst = 0;
    for (int it = 0; it < trials; ++it) {
        cf v[R];
        cld in[R];
        for (int n = 0; n < R; ++n) {
            v[n] = cf{rnd(), rnd()};
            in[n] = to_ld(v[n]);
        }
        rpf::SmallDft<R>::run(v);
        long double rms = 0;
        cld ex[R];
        for (int k = 0; k < R; ++k) {
            cld x = 0;
            for (int n = 0; n < R; ++n) x += in[n] * wexact(static_cast<long>(n) * k, R);
            ex[k] = x;
            rms += std::norm(x);
        }
        rms = sqrtl(rms / R);
        for (int k = 0; k < R; ++k) {
            const long double e = std::abs(to_ld(v[k]) - ex[k]);
            e2 += e * e;
            worst = std::max(worst, e / rms);
        }
        o2 += rms * rms * R;
    }
    out[0] = static_cast<double>(sqrtl(e2 / o2));
    out[1] = static_cast<double>(worst);
}
}  // namespace
extern "C" int rpf_analysis_radix_error(int R, int trials, double* out)
{
    switch (R) {
#define RC(r) case r: radix_error<r>(trials, out); return 0
        RC(2); RC(3); RC(4); RC(5); RC(6); RC(7); RC(8); RC(9); RC(10); RC(11); RC(12); RC(13); RC(14); RC(15); RC(16);
        RC(17); RC(18); RC(19); RC(20); RC(21); RC(22); RC(23); RC(24); RC(25);
#undef RC
    }
    return -1;
}

// The last pass beside a line: outputs = one line of amplitude A at k0 and unit complex noise elsewhere; inputs = the exact
// inverse transform rounded to float.  rms error of SmallDft<R> on the WEAK outputs against the exact transform of those
// float inputs (the butterfly's own arithmetic), in units of eps * A; out[1]: the same for the exact transform of the
// UNROUNDED inputs (what the rounding of the inputs alone costs -- no butterfly can avoid that).
namespace {
template <int R>
void line_exposure(int trials, double A, double* out)
{
    unsigned long long s = 0xD1B54A32D192ED03ull * (R + 7);
    auto uni = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (s >> 11) * (1.0L / 9007199254740992.0L);
    };
    auto gauss = [&]() { return sqrtl(-2.0L * logl(uni() + 1e-300L)) * cosl(kTwoPi * uni()); };
    long double e_arith = 0, e_quant = 0;
    long cnt = 0;
    for (int it = 0; it < trials; ++it) {
        const int k0 = it % R;
        cld want[R], in[R], inq[R];
        for (int k = 0; k < R; ++k) want[k] = cld(gauss(), gauss()) * 0.70710678L;
        const ld ph = kTwoPi * uni();
        want[k0] = cld(A * cosl(ph), A * sinl(ph));
        cf v[R];
        for (int n = 0; n < R; ++n) {
            cld x = 0;
            for (int k = 0; k < R; ++k) x += want[k] * std::conj(wexact(static_cast<long>(n) * k, R));
            in[n] = x / static_cast<ld>(R);
            v[n] = to_cf(in[n]);
            inq[n] = to_ld(v[n]);
        }
        rpf::SmallDft<R>::run(v);
        for (int k = 0; k < R; ++k) {
            if (k == k0) continue;
            cld xq = 0;
            for (int n = 0; n < R; ++n) xq += inq[n] * wexact(static_cast<long>(n) * k, R);
            e_arith += std::norm(to_ld(v[k]) - xq);
            e_quant += std::norm(xq - want[k]);
            ++cnt;
        }
    }
    const long double eps = 5.9604644775390625e-08L;
    out[0] = static_cast<double>(sqrtl(e_arith / cnt) / (eps * A));
    out[1] = static_cast<double>(sqrtl(e_quant / cnt) / (eps * A));
}
}  // namespace
extern "C" int rpf_analysis_line_exposure(int R, int trials, double A, double* out)
{
    switch (R) {
#define RC(r) case r: line_exposure<r>(trials, A, out); return 0
        RC(2); RC(3); RC(4); RC(5); RC(6); RC(7); RC(8); RC(9); RC(10); RC(11); RC(12); RC(13); RC(14); RC(15); RC(16);
        RC(17); RC(18); RC(19); RC(20); RC(21); RC(22); RC(23); RC(24); RC(25);
#undef RC
    }
    return -1;
}
