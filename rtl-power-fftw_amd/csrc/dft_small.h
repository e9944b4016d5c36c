// dft_small.h -- in-register DFTs of the small composite lengths the mixed-radix kernel
// (rpf_mixed.hip) uses as radices: every length from 2 to 25.
//
// SmallDft<R>::run(v): v[k] <- sum_n v[n] W_R^{n k}, W_R = e^{-2 pi i / R}, natural order in and
// out, everything on registers (all indices are compile-time constants after unrolling, so the
// index maps below are register renaming, not data movement).
//   * coprime factors (6 = 2x3, 10 = 2x5, 12 = 4x3, 15 = 3x5, 18 = 2x9, 20 = 4x5, 24 = 8x3): Good-Thomas prime-factor
//     map -- no twiddles between the two stages at all;
//   * 9 = 3x3, 25 = 5x5: one Cooley-Tukey step with constant twiddles W_R^m (correctly rounded
//     floats from a constexpr double evaluation);
//   * the odd primes 7, 11, 13, 17, 19, 23: the symmetric direct form (PrimeDft), and with it 14, 21, 22;
//   * 2, 4, 8, 16: fft_core.h's butterflies (K1's).
// Like fft_core.h this is plain C++17 on `cf` so that the host emulator under tests/emul checks
// every radix against a naive double DFT on a machine without a GPU.
#pragma once

#include "fft_core.h"

namespace rpf {

// ---- constexpr cos / sin of a fraction of a turn (double; the float rounding of the result is
// what the kernels use).  |angle| <= pi/4 after the octant reduction, 12 Taylor terms.
constexpr double kPiD = 3.14159265358979323846264338327950288;

constexpr double taylor_sin(double x)
{
    double term = x, sum = x;
    for (int k = 1; k < 14; ++k) {
        term *= -x * x / ((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double taylor_cos(double x)
{
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 14; ++k) {
        term *= -x * x / ((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi num / den, 0 <= num < den
constexpr double cos_turn(int num, int den)
{
    // reduce to the first octant: angle = 2 pi num/den
    const int n8 = 8 * num;                  // angle in units of (2 pi / 8) / den
    const int oct = n8 / den;                // 0..7
    const double x = 2.0 * kPiD * num / den;
    switch (oct) {
        case 0: return taylor_cos(x);
        case 1: return taylor_sin(kPiD / 2 - x);
        case 2: return -taylor_sin(x - kPiD / 2);
        case 3: return -taylor_cos(kPiD - x);
        case 4: return -taylor_cos(x - kPiD);
        case 5: return -taylor_sin(3 * kPiD / 2 - x);
        case 6: return taylor_sin(x - 3 * kPiD / 2);
        default: return taylor_cos(2 * kPiD - x);
    }
}
constexpr double sin_turn(int num, int den)
{
    const int oct = 8 * num / den;
    const double x = 2.0 * kPiD * num / den;
    switch (oct) {
        case 0: return taylor_sin(x);
        case 1: return taylor_cos(kPiD / 2 - x);
        case 2: return taylor_cos(x - kPiD / 2);
        case 3: return taylor_sin(kPiD - x);
        case 4: return -taylor_sin(x - kPiD);
        case 5: return -taylor_cos(3 * kPiD / 2 - x);
        case 6: return -taylor_cos(x - 3 * kPiD / 2);
        default: return -taylor_sin(2 * kPiD - x);
    }
}

// complex product with a compile-time constant w: the constant rides in a scalar register pair
// (VOP3P takes one scalar source), so it costs neither vector registers nor v_mov instructions.
RPF_HD cf cmul_k(cf a, cf w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "s"(w));
    return d;
#else
    return cmul(a, w);
#endif
}

// acc + a w, w as in cmul_k: two fused multiply-adds -- two roundings per component where cmul_k + add has three
RPF_HD cf cmac_k(cf acc, cf a, cf w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    cf d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "s"(w), "v"(acc));
    return d;
#else
    return cf{__builtin_fmaf(-a.y, w.y, __builtin_fmaf(a.x, w.x, acc.x)), __builtin_fmaf(a.x, w.y, __builtin_fmaf(a.y, w.x, acc.y))};
#endif
}

// a * W_R^M, M a compile-time constant
template <int R, int M>
RPF_HD cf mul_wconst(cf a)
{
    constexpr int m = ((M % R) + R) % R;
    if constexpr (m == 0) return a;
    else if constexpr (4 * m == R) return mul_mi(a);
    else if constexpr (2 * m == R) return -a;
    else if constexpr (4 * m == 3 * R) return mul_pi(a);
    else {
        constexpr float c = static_cast<float>(cos_turn(m, R));
        constexpr float s = static_cast<float>(-sin_turn(m, R));
        return cmul_k(a, cf{c, s});
    }
}

template <int R>
struct SmallDft;

template <>
struct SmallDft<2> {
    static RPF_HD void run(cf* v) { Dft<2>::run(v); }
};
template <>
struct SmallDft<4> {
    static RPF_HD void run(cf* v) { Dft<4>::run(v); }
};
template <>
struct SmallDft<8> {
    static RPF_HD void run(cf* v) { Dft<8>::run(v); }
};
template <>
struct SmallDft<16> {
    static RPF_HD void run(cf* v) { Dft<16>::run(v); }
};

template <>
struct SmallDft<3> {
    static RPF_HD void run(cf* v)
    {
        constexpr float kS3 = static_cast<float>(sin_turn(1, 3));
        const cf s = v[1] + v[2], d = v[1] - v[2];
        const cf m = v[0] - s * 0.5f;                     // v0 + cos(2 pi/3) (v1 + v2)
        const cf jd = mul_mi(d) * kS3;                    // -i sin(2 pi/3) (v1 - v2)
        v[0] = v[0] + s;
        v[1] = m + jd;
        v[2] = m - jd;
    }
};

template <>
struct SmallDft<5> {
    static RPF_HD void run(cf* v)
    {
        constexpr float kC1 = static_cast<float>(cos_turn(1, 5)), kC2 = static_cast<float>(cos_turn(2, 5));
        constexpr float kS1 = static_cast<float>(sin_turn(1, 5)), kS2 = static_cast<float>(sin_turn(2, 5));
        const cf s14 = v[1] + v[4], d14 = v[1] - v[4];
        const cf s23 = v[2] + v[3], d23 = v[2] - v[3];
        const cf a1 = v[0] + s14 * kC1 + s23 * kC2;
        const cf a2 = v[0] + s14 * kC2 + s23 * kC1;
        const cf b1 = mul_mi(d14 * kS1 + d23 * kS2);      // -i (...)
        const cf b2 = mul_mi(d14 * kS2 - d23 * kS1);
        v[0] = v[0] + s14 + s23;
        v[1] = a1 + b1;
        v[4] = a1 - b1;
        v[2] = a2 + b2;
        v[3] = a2 - b2;
    }
};

// Odd prime P by the symmetric direct form: with s_j = x_j + x_{P-j}, d_j = x_j - x_{P-j} (j <= h = (P-1)/2),
//   X_0 = x_0 + sum_j s_j,   X_k, X_{P-k} = a_k -+ i b_k,   a_k = x_0 + sum_j s_j cos(2 pi jk/P),  b_k = sum_j d_j sin(2 pi jk/P)
// -- (P-1)^2 / 2 packed multiply-adds with real constants, about 4 instructions per point at P = 7 (like the
// hand-written 5 above), 7 at P = 13, 12 at P = 23.
template <int P>
struct PrimeTable {
    float c[P], s[P];
    constexpr PrimeTable() : c(), s()
    {
        for (int m = 0; m < P; ++m) {
            c[m] = static_cast<float>(cos_turn(m, P));
            s[m] = static_cast<float>(sin_turn(m, P));
        }
    }
};
template <int P>
struct PrimeDft {
    static constexpr PrimeTable<P> tab{};
    static RPF_HD void run(cf* v)
    {
        constexpr int H = (P - 1) / 2;
        cf s[H], d[H];
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            s[j - 1] = v[j] + v[P - j];
            d[j - 1] = v[j] - v[P - j];
        }
        const cf x0 = v[0];
        cf sum = x0;
#pragma unroll
        for (int j = 0; j < H; ++j) sum = sum + s[j];
        v[0] = sum;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            cf a = x0, b = cf{0.0f, 0.0f};
#pragma unroll
            for (int j = 1; j <= H; ++j) {
                a = a + s[j - 1] * tab.c[(j * k) % P];
                if (j == 1) b = d[0] * tab.s[k % P];
                else b = b + d[j - 1] * tab.s[(j * k) % P];
            }
            v[k] = add_mi(a, b);          // a - i b
            v[P - k] = sub_mi(a, b);      // a + i b
        }
    }
};

constexpr int mod_inverse(int a, int m)      // a^-1 mod m (a, m coprime, small)
{
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 1;
}

// R = A * B, gcd(A, B) = 1 (Good-Thomas): n = (B n1 + A n2) mod R, k = (B b k1 + A a k2) mod R with
// b = B^-1 mod A, a = A^-1 mod B; X[k] = sum_{n1,n2} x[n] W_A^{n1 k1} W_B^{n2 k2}.
template <int R, int A, int B>
struct PfaDft {
    static RPF_HD void run(cf* v)
    {
        static_assert(R == A * B, "");
        constexpr int bi = mod_inverse(B % A, A), ai = mod_inverse(A % B, B);
        cf t[R];
        cf u[A];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) u[n1] = v[(B * n1 + A * n2) % R];
            SmallDft<A>::run(u);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) t[k1 * B + n2] = u[k1];
        }
        cf w[B];
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) w[n2] = t[k1 * B + n2];
            SmallDft<B>::run(w);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) v[(B * bi * k1 + A * ai * k2) % R] = w[k2];
        }
    }
};

// R = A * B by one Cooley-Tukey step: n = B n1 + n2, k = k1 + A k2, twiddle W_R^{n2 k1}.
template <int R, int A, int B>
struct CtDft {
    template <int IDX>
    static RPF_HD void twiddle(cf* t, std::integral_constant<int, IDX>)
    {
        if constexpr (IDX < R) {
            constexpr int k1 = IDX / B, n2 = IDX % B;
            t[IDX] = mul_wconst<R, k1 * n2>(t[IDX]);
            twiddle(t, std::integral_constant<int, IDX + 1>{});
        }
    }
    static RPF_HD void run(cf* v)
    {
        static_assert(R == A * B, "");
        cf t[R];
        cf u[A];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) u[n1] = v[B * n1 + n2];
            SmallDft<A>::run(u);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) t[k1 * B + n2] = u[k1];
        }
        twiddle(t, std::integral_constant<int, 0>{});
        cf w[B];
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) w[n2] = t[k1 * B + n2];
            SmallDft<B>::run(w);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = w[k2];
        }
    }
};

template <>
struct SmallDft<7> {
    static RPF_HD void run(cf* v) { PrimeDft<7>::run(v); }
};
template <>
struct SmallDft<11> {
    static RPF_HD void run(cf* v) { PrimeDft<11>::run(v); }
};
template <>
struct SmallDft<13> {
    static RPF_HD void run(cf* v) { PrimeDft<13>::run(v); }
};
template <>
struct SmallDft<17> {
    static RPF_HD void run(cf* v) { PrimeDft<17>::run(v); }
};
template <>
struct SmallDft<19> {
    static RPF_HD void run(cf* v) { PrimeDft<19>::run(v); }
};
template <>
struct SmallDft<23> {
    static RPF_HD void run(cf* v) { PrimeDft<23>::run(v); }
};
template <>
struct SmallDft<14> {
    static RPF_HD void run(cf* v) { PfaDft<14, 2, 7>::run(v); }
};
template <>
struct SmallDft<21> {
    static RPF_HD void run(cf* v) { PfaDft<21, 3, 7>::run(v); }
};
template <>
struct SmallDft<22> {
    static RPF_HD void run(cf* v) { PfaDft<22, 2, 11>::run(v); }
};
template <>
struct SmallDft<6> {
    static RPF_HD void run(cf* v) { PfaDft<6, 2, 3>::run(v); }
};
template <>
struct SmallDft<10> {
    static RPF_HD void run(cf* v) { PfaDft<10, 2, 5>::run(v); }
};
template <>
struct SmallDft<12> {
    static RPF_HD void run(cf* v) { PfaDft<12, 4, 3>::run(v); }
};
template <>
struct SmallDft<15> {
    static RPF_HD void run(cf* v) { PfaDft<15, 3, 5>::run(v); }
};
template <>
struct SmallDft<20> {
    static RPF_HD void run(cf* v) { PfaDft<20, 4, 5>::run(v); }
};
template <>
struct SmallDft<9>;
template <>
struct SmallDft<18> {
    static RPF_HD void run(cf* v);
};
template <>
struct SmallDft<24> {
    static RPF_HD void run(cf* v) { PfaDft<24, 8, 3>::run(v); }
};
template <>
struct SmallDft<9> {
    static RPF_HD void run(cf* v) { CtDft<9, 3, 3>::run(v); }
};
RPF_HD void SmallDft<18>::run(cf* v) { PfaDft<18, 2, 9>::run(v); }
template <>
struct SmallDft<25> {
    static RPF_HD void run(cf* v) { CtDft<25, 5, 5>::run(v); }
};

}  // namespace rpf
