// bluestein_tables.h -- host-side tables for transform lengths that are not a
// power of two (the reference accepts any even N because FFTW does,
// /root/reference/src/params.cxx:150-155, datastore.cxx:32; the man page's own
// example uses -b 500).  Bluestein's identity  nk = (n^2 + k^2 - (k-n)^2)/2  turns
// the N-point DFT into a length-M circular convolution, M a power of two >= 2N-1:
//
//   X[k] = conj(b[k]) * sum_n ( x[n] conj(b[n]) ) * b[k-n],     b[m] = exp(+i pi m^2 / N)
//
// Only |X[k]|^2 is needed, so the final conj(b[k]) factor drops out.  Tables
// (evaluated in long double, phases reduced exactly with m^2 mod 2N):
//   g[n]    = (-1)^n * window[n] * conj(b[n])      n < N   (unpack multiplier)
//   bhat[j] = FFT_M(b extended circularly)[j] / M   j < M   (frequency-domain kernel)
// Plain C++ (used by the engine and by the test emulator).
#pragma once

#include <cmath>
#include <complex>
#include <vector>

namespace rpf {

inline int bluestein_length(int N)
{
    int M = 64;
    while (M < 2 * N - 1) M *= 2;
    return M;
}

namespace detail {
inline void fft_pow2(std::vector<std::complex<long double>>& a)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    // W_n^j, j < n/2, evaluated once (the generic path plans lengths up to 2^21)
    const long double pi = 3.141592653589793238462643383279502884L;
    std::vector<std::complex<long double>> w(n / 2 ? n / 2 : 1);
    for (size_t j = 0; j < n / 2; ++j) {
        const long double ang = -2 * pi * static_cast<long double>(j) / static_cast<long double>(n);
        w[j] = std::complex<long double>(std::cos(ang), std::sin(ang));
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<long double> u = a[i + k], v = a[i + k + len / 2] * w[k * step];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}
}  // namespace detail

// g: N complex (re,im interleaved floats); bhat: M complex.  window may be null.
inline void make_bluestein_tables(int N, const float* window, std::vector<float>& g, std::vector<float>& bhat)
{
    const int M = bluestein_length(N);
    const long double pi = 3.141592653589793238462643383279502884L;
    auto chirp = [&](long long m) {      // b[m] = exp(+i pi m^2 / N), phase reduced exactly
        const long long r = (m * m) % (2LL * N);
        const long double ang = pi * static_cast<long double>(r) / static_cast<long double>(N);
        return std::complex<long double>(std::cos(ang), std::sin(ang));
    };
    g.resize(2 * static_cast<size_t>(N));
    for (int n = 0; n < N; ++n) {
        const std::complex<long double> c = std::conj(chirp(n));
        // the window value is the float the reference multiplies by; the sign flip is exact
        const long double w = (window ? static_cast<long double>(window[n]) : 1.0L) * ((n & 1) ? -1.0L : 1.0L);
        g[2 * n] = static_cast<float>(w * c.real());
        g[2 * n + 1] = static_cast<float>(w * c.imag());
    }
    std::vector<std::complex<long double>> b(M, std::complex<long double>(0, 0));
    b[0] = chirp(0);
    for (int m = 1; m < N; ++m) b[m] = b[M - m] = chirp(m);
    detail::fft_pow2(b);
    bhat.resize(2 * static_cast<size_t>(M));
    for (int j = 0; j < M; ++j) {
        bhat[2 * j] = static_cast<float>(b[j].real() / M);
        bhat[2 * j + 1] = static_cast<float>(b[j].imag() / M);
    }
}

}  // namespace rpf
