// dft_small_wide.h -- the small DFTs of dft_small.h once more, in double: WideDft<R>::run(v), v[k] <- sum_n v[n] W_R^{nk},
// R = 2 ... 25, natural order in and out, on registers.
//
// Why (tools/analysis/parity_passes.cpp): with deterministic lines 1e4 above the weakest bins, what a float32
// transform loses is lost in its LAST pass -- there the line's energy has collected in the R_last inputs of ONE butterfly,
// every rounding inside that butterfly is relative to the line and lands on the R_last - 1 weak bins that share it
// (measured: 0.19 - 0.31 eps x line amplitude for every radix >= 3, against 0.08 - 0.10 for the rounding of the butterfly's
// float inputs, which nothing can avoid).  The split / paired forms at 42000 ... 108000 bins sat 0.7 - 1.1e-6 from float64
// truth where the CPU path -- whose odd-radix butterflies accumulate in double -- has 0.3 - 0.5e-6; with the last pass
// (and only it) in double they sit at 0.3e-6.  The squares go straight into the f64 accumulators
// (mix_last_pass_accumulate), so the two v_cvt_f64_f32 per bin of the float path move in front of the butterfly.
// Structure as dft_small.h: Good-Thomas for coprime factors, one Cooley-Tukey step for 4, 8, 9, 16, 25, the symmetric
// direct form for the odd primes (3 and 5 included).  Plain C++17: compiled by hipcc and by the host emulator.
#pragma once

#include "dft_small.h"

namespace rpf {

typedef double cd __attribute__((ext_vector_type(2)));

RPF_HD cd wide_mul_mi(cd a) { return cd{a.y, -a.x}; }      // -i a
RPF_HD cd wide_mul_pi(cd a) { return cd{-a.y, a.x}; }      //  i a
RPF_HD cd wide_cmul(cd a, double c, double s)              // a (c + i s)
{
    return cd{__builtin_fma(-a.y, s, a.x * c), __builtin_fma(a.x, s, a.y * c)};
}
// a * W_R^M, M a compile-time constant
template <int R, int M>
RPF_HD cd wide_mul_wconst(cd a)
{
    constexpr int m = ((M % R) + R) % R;
    if constexpr (m == 0) return a;
    else if constexpr (4 * m == R) return wide_mul_mi(a);
    else if constexpr (2 * m == R) return -a;
    else if constexpr (4 * m == 3 * R) return wide_mul_pi(a);
    else return wide_cmul(a, cos_turn(m, R), -sin_turn(m, R));
}

template <int R>
struct WideDft;

template <>
struct WideDft<2> {
    static RPF_HD void run(cd* v)
    {
        const cd a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <int P>
struct WidePrimeTable {
    double c[P], s[P];
    constexpr WidePrimeTable() : c(), s()
    {
        for (int m = 0; m < P; ++m) {
            c[m] = cos_turn(m, P);
            s[m] = sin_turn(m, P);
        }
    }
};
// odd prime P: s_j = x_j + x_{P-j}, d_j = x_j - x_{P-j};  X_k, X_{P-k} = (x_0 + sum_j s_j cos) -+ i sum_j d_j sin
template <int P>
struct WidePrimeDft {
    static constexpr WidePrimeTable<P> tab{};
    static RPF_HD void run(cd* v)
    {
        constexpr int H = (P - 1) / 2;
        cd s[H], d[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            s[j - 1] = v[j] + v[P - j];
            d[j - 1] = v[j] - v[P - j];
        }
        const cd x0 = v[0];
        cd sum = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) sum = sum + s[j];
        v[0] = sum;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            cd a = x0, b = cd{0.0, 0.0};
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                const double c = tab.c[(j * k) % P], sn = tab.s[(j * k) % P];
                a = cd{__builtin_fma(s[j - 1].x, c, a.x), __builtin_fma(s[j - 1].y, c, a.y)};
                b = cd{__builtin_fma(d[j - 1].x, sn, b.x), __builtin_fma(d[j - 1].y, sn, b.y)};
            }
            v[k] = cd{a.x + b.y, a.y - b.x};          // a - i b
            v[P - k] = cd{a.x - b.y, a.y + b.x};      // a + i b
        }
    }
};

// R = A B, gcd(A, B) = 1 (Good-Thomas, index maps as PfaDft)
template <int R, int A, int B>
struct WidePfaDft {
    static RPF_HD void run(cd* v)
    {
        static_assert(R == A * B, "");
        constexpr int bi = mod_inverse(B % A, A), ai = mod_inverse(A % B, B);
        cd t[R];
        cd u[A];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) u[n1] = v[(B * n1 + A * n2) % R];
            WideDft<A>::run(u);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) t[k1 * B + n2] = u[k1];
        }
        cd w[B];
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) w[n2] = t[k1 * B + n2];
            WideDft<B>::run(w);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) v[(B * bi * k1 + A * ai * k2) % R] = w[k2];
        }
    }
};

// R = A B by one Cooley-Tukey step: n = B n1 + n2, k = k1 + A k2, twiddle W_R^{n2 k1}
template <int R, int A, int B>
struct WideCtDft {
    template <int IDX>
    static RPF_HD void twiddle(cd* t, std::integral_constant<int, IDX>)
    {
        if constexpr (IDX < R) {
            constexpr int k1 = IDX / B, n2 = IDX % B;
            t[IDX] = wide_mul_wconst<R, k1 * n2>(t[IDX]);
            twiddle(t, std::integral_constant<int, IDX + 1>{});
        }
    }
    static RPF_HD void run(cd* v)
    {
        static_assert(R == A * B, "");
        cd t[R];
        cd u[A];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) u[n1] = v[B * n1 + n2];
            WideDft<A>::run(u);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) t[k1 * B + n2] = u[k1];
        }
        twiddle(t, std::integral_constant<int, 0>{});
        cd w[B];
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) w[n2] = t[k1 * B + n2];
            WideDft<B>::run(w);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = w[k2];
        }
    }
};

#define RPF_WIDE_DFT(R, ...)                                 \
    template <>                                              \
    struct WideDft<R> {                                      \
        static RPF_HD void run(cd* v) { __VA_ARGS__::run(v); } \
    }
RPF_WIDE_DFT(3, WidePrimeDft<3>);
RPF_WIDE_DFT(5, WidePrimeDft<5>);
RPF_WIDE_DFT(7, WidePrimeDft<7>);
RPF_WIDE_DFT(11, WidePrimeDft<11>);
RPF_WIDE_DFT(13, WidePrimeDft<13>);
RPF_WIDE_DFT(17, WidePrimeDft<17>);
RPF_WIDE_DFT(19, WidePrimeDft<19>);
RPF_WIDE_DFT(23, WidePrimeDft<23>);
RPF_WIDE_DFT(4, WideCtDft<4, 2, 2>);
RPF_WIDE_DFT(8, WideCtDft<8, 4, 2>);
RPF_WIDE_DFT(9, WideCtDft<9, 3, 3>);
RPF_WIDE_DFT(16, WideCtDft<16, 4, 4>);
RPF_WIDE_DFT(25, WideCtDft<25, 5, 5>);
RPF_WIDE_DFT(6, WidePfaDft<6, 2, 3>);
RPF_WIDE_DFT(10, WidePfaDft<10, 2, 5>);
RPF_WIDE_DFT(12, WidePfaDft<12, 4, 3>);
RPF_WIDE_DFT(14, WidePfaDft<14, 2, 7>);
RPF_WIDE_DFT(15, WidePfaDft<15, 3, 5>);
RPF_WIDE_DFT(18, WidePfaDft<18, 2, 9>);
RPF_WIDE_DFT(20, WidePfaDft<20, 4, 5>);
RPF_WIDE_DFT(21, WidePfaDft<21, 3, 7>);
RPF_WIDE_DFT(22, WidePfaDft<22, 2, 11>);
RPF_WIDE_DFT(24, WidePfaDft<24, 8, 3>);
#undef RPF_WIDE_DFT

// ---- the wide LAST pass as one piece: float values in, |X|^2 folded into the f64 accumulators, nothing in between kept
// longer or wider than it has to be.  The kernels that run it are the register-bound ones (nine-wave workgroups, 168
// VGPRs): a butterfly that converts all R inputs up front and holds R double outputs needs 4 R registers twice over and
// spills -- and the spills, not the f64 arithmetic, were what the wide pass cost (21000 bins: 117 scratch instructions
// per thread against 32, 4.7 x the frame's own bytes in scratch traffic).  Here, for a composite R = A B: the level-1
// butterflies convert their A inputs as they load them, their outputs wait for level 2 as `wide_mid`s, and each level-2
// butterfly's B outputs go straight into the accumulators.
// RPF_WIDE_MID_FLOAT=1 (measured, NOT shipped): the values between the two levels as floats -- 2 R registers instead of
// 4 R, half of the wide pass's extra spills gone and up to 2 x of its cost back (21000: 150 -> 244, 98304: 79 -> 161
// Gsample/s; median + 5 %) -- but that one more float rounding sits where the line is already concentrated, and three
// sizes (75000, 88000, 108000) went back over the parity bar on a held-out stream.  Shipped: doubles between the levels.
#ifndef RPF_WIDE_MID_FLOAT
#define RPF_WIDE_MID_FLOAT 0
#endif
#if RPF_WIDE_MID_FLOAT
typedef cf wide_mid;
RPF_HD wide_mid to_mid(cd a) { return cf{static_cast<float>(a.x), static_cast<float>(a.y)}; }
RPF_HD cd from_mid(wide_mid a) { return cd{static_cast<double>(a.x), static_cast<double>(a.y)}; }
#else
typedef cd wide_mid;
RPF_HD wide_mid to_mid(cd a) { return a; }
RPF_HD cd from_mid(wide_mid a) { return a; }
#endif
RPF_HD cd widen(cf a) { return cd{static_cast<double>(a.x), static_cast<double>(a.y)}; }
RPF_HD void fold_square(double& acc, cd x) { acc = __builtin_fma(x.y, x.y, __builtin_fma(x.x, x.x, acc)); }

// Out: what becomes of output k -- the last pass folds |X_k|^2 into acc[k]; the pass before it (kWideLastTwo) multiplies
// by the twiddle and rounds back to float.
struct WideFold {
    double* acc;
    RPF_HD void operator()(int k, cd x) const { fold_square(acc[k], x); }
};
struct WideTwiddleStore {
    cf* out;
    const cf* tw;          // tw[k - 1] for k >= 1
    RPF_HD void operator()(int k, cd x) const
    {
        if (k > 0) x = wide_cmul(x, static_cast<double>(tw[k - 1].x), static_cast<double>(tw[k - 1].y));
        out[k] = cf{static_cast<float>(x.x), static_cast<float>(x.y)};
    }
};

template <int R, int A, int B, bool PFA>
struct WideTwoLevel {
    // level 1, column n2: A inputs converted as they are loaded, A outputs (times W_R^{n2 k1} in the Cooley-Tukey form)
    template <int N2>
    static RPF_HD void column(const cf* v, wide_mid* t, std::integral_constant<int, N2>)
    {
        if constexpr (N2 < B) {
            cd u[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) u[n1] = widen(v[PFA ? (B * n1 + A * N2) % R : B * n1 + N2]);
            WideDft<A>::run(u);
            store(u, t, std::integral_constant<int, N2>{}, std::integral_constant<int, 0>{});
            column(v, t, std::integral_constant<int, N2 + 1>{});
        }
    }
    template <int N2, int K1>
    static RPF_HD void store(const cd* u, wide_mid* t, std::integral_constant<int, N2>, std::integral_constant<int, K1>)
    {
        if constexpr (K1 < A) {
            if constexpr (PFA) t[K1 * B + N2] = to_mid(u[K1]);
            else t[K1 * B + N2] = to_mid(wide_mul_wconst<R, K1 * N2>(u[K1]));
            store(u, t, std::integral_constant<int, N2>{}, std::integral_constant<int, K1 + 1>{});
        }
    }
    template <class Out>
    static RPF_HD void run(const cf* v, const Out& out)
    {
        static_assert(R == A * B, "");
        constexpr int bi = mod_inverse(B % A, A), ai = mod_inverse(A % B, B);
        wide_mid t[R];
        column(v, t, std::integral_constant<int, 0>{});
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            cd w[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) w[n2] = from_mid(t[k1 * B + n2]);
            WideDft<B>::run(w);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) out(PFA ? (B * bi * k1 + A * ai * k2) % R : k1 + A * k2, w[k2]);
        }
    }
};

// v[0 .. R): the pass's float inputs (left untouched); out(k, X_k) for every output
template <int R>
struct WidePass {
    template <class Out>
    static RPF_HD void run(const cf* v, const Out& out)
    {
        if constexpr (R == 6) WideTwoLevel<6, 2, 3, true>::run(v, out);
        else if constexpr (R == 10) WideTwoLevel<10, 2, 5, true>::run(v, out);
        else if constexpr (R == 12) WideTwoLevel<12, 4, 3, true>::run(v, out);
        else if constexpr (R == 14) WideTwoLevel<14, 2, 7, true>::run(v, out);
        else if constexpr (R == 15) WideTwoLevel<15, 3, 5, true>::run(v, out);
        else if constexpr (R == 18) WideTwoLevel<18, 2, 9, true>::run(v, out);
        else if constexpr (R == 20) WideTwoLevel<20, 4, 5, true>::run(v, out);
        else if constexpr (R == 21) WideTwoLevel<21, 3, 7, true>::run(v, out);
        else if constexpr (R == 22) WideTwoLevel<22, 2, 11, true>::run(v, out);
        else if constexpr (R == 24) WideTwoLevel<24, 8, 3, true>::run(v, out);
        else if constexpr (R == 8) WideTwoLevel<8, 4, 2, false>::run(v, out);
        else if constexpr (R == 9) WideTwoLevel<9, 3, 3, false>::run(v, out);
        else if constexpr (R == 16) WideTwoLevel<16, 4, 4, false>::run(v, out);
        else if constexpr (R == 25) WideTwoLevel<25, 5, 5, false>::run(v, out);
        else {
            // small radices and the odd primes: one level
            cd w[R];
#pragma unroll
            for (int n = 0; n < R; ++n) w[n] = widen(v[n]);
            WideDft<R>::run(w);
#pragma unroll
            for (int k = 0; k < R; ++k) out(k, w[k]);
        }
    }
};

}  // namespace rpf
