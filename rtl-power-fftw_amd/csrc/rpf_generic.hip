// rpf_generic.hip -- the catch-all path: every even N that none of the tuned kernel
// families covers (powers of two above 262144, other even N above 131072).
//
// The reference takes any even N because FFTW does (/root/reference/src/params.cxx:150-155,
// datastore.cxx:32); the tuned families (rpf_kernels.hip: one workgroup's LDS; rpf_fourstep.hip:
// two LDS-sized factors) stop at 262144 bins.  Beyond that the transform runs as a plain
// Stockham autosort FFT through HBM -- radix-4 passes (one radix-2 pass first when log2 M is
// odd), natural order in and out, one launch per pass over a batch of frames -- which is
// slow next to the fused kernels (every pass moves 16 bytes per sample) but has no size limit
// other than memory, and is exact to the same float32 bar.  Lengths that are not powers of two
// go through Bluestein's identity (bluestein_tables.h) on top of it.
//
//   gen_load      u8 IQ -> (v - 127) * (-1)^n [* window]  (Bluestein: * g[n], zero-padded to M)
//   gen_radix4/2  y[q + s (R p + r)] = W_n^{p r} * sum_k x[q + s (p + k n/R)] W_R^{k r}
//   gen_mul_conj  Bluestein: z = conj(A * bhat)
//   gen_accum     pwr[k] += sum over the batch of |X[k]|^2, in double, one writer per bin
//
// Twiddles W_M^i are read from a two-level table, W_M^i = T1[i >> h] * T0[i & (2^h - 1)]
// (2 * sqrt(M) entries instead of M; one extra complex product, both factors correctly rounded).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "bluestein_tables.h"
#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

constexpr int kThreads = 256;

struct TwoLevel {
    const cf* t0;      // W_M^j, j < 2^h
    const cf* t1;      // W_M^{j 2^h}
    int h;
    unsigned mask;
};

__device__ __forceinline__ cf tw_lookup(const TwoLevel& t, unsigned idx)
{
    return cmul(t.t1[idx >> t.h], t.t0[idx & t.mask]);
}

// BLU: mult = g (N complex: (-1)^n, window and chirp folded in); else window (N floats) or null.
template <bool BLU>
__global__ __launch_bounds__(kThreads) void gen_load_kernel(const uint8_t* __restrict__ stream, int N, int M,
                                                           long total, const float* __restrict__ window,
                                                           const cf* __restrict__ g, cf* __restrict__ out)
{
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= total) return;
    const long f = i / M;
    const int n = static_cast<int>(i - f * M);
    cf v = cf{0.0f, 0.0f};
    if (n < N) {
        const uint32_t iq = *reinterpret_cast<const uint16_t*>(stream + (f * N + n) * 2);
        const cf x = iq_plus_2p23(iq) - (kTwo23 + 127.0f);            // exact (datastore.cxx:75)
        if constexpr (BLU) {
            v = cmul(x, g[n]);
        } else {
            const float s = (n & 1) ? -1.0f : 1.0f;                    // datastore.cxx:73,76-77
            v = window ? x * (window[n] * s) : x * s;
        }
    }
    out[i] = v;
}

__global__ __launch_bounds__(kThreads) void gen_radix2_kernel(const cf* __restrict__ x, cf* __restrict__ y, int M,
                                                             int ls, long total, TwoLevel tw)
{
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= total) return;
    const int half = M >> 1;
    const long f = i / half;
    const unsigned j = static_cast<unsigned>(i - f * half);
    const unsigned s = 1u << ls, p = j >> ls, q = j & (s - 1);
    const unsigned m = static_cast<unsigned>(M) >> (ls + 1);          // n / 2, n = M / s
    const cf* xf = x + f * M;
    cf* yf = y + f * M;
    const cf a = xf[q + s * p], b = xf[q + s * (p + m)];
    yf[q + s * (2 * p)] = a + b;
    yf[q + s * (2 * p + 1)] = cmul(a - b, tw_lookup(tw, p << ls));
}

__global__ __launch_bounds__(kThreads) void gen_radix4_kernel(const cf* __restrict__ x, cf* __restrict__ y, int M,
                                                             int ls, long total, TwoLevel tw)
{
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= total) return;
    const int quarter = M >> 2;
    const long f = i / quarter;
    const unsigned j = static_cast<unsigned>(i - f * quarter);
    const unsigned s = 1u << ls, p = j >> ls, q = j & (s - 1);
    const unsigned n1 = static_cast<unsigned>(M) >> (ls + 2);         // n / 4, n = M / s
    const cf* xf = x + f * M;
    cf* yf = y + f * M;
    const cf a = xf[q + s * p], b = xf[q + s * (p + n1)], c = xf[q + s * (p + 2 * n1)], d = xf[q + s * (p + 3 * n1)];
    const cf apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
    const unsigned i1 = p << ls;                                        // W_n^p = W_M^{p s}
    yf[q + s * (4 * p)] = apc + bpd;
    yf[q + s * (4 * p + 1)] = cmul(add_mi(amc, bmd), tw_lookup(tw, i1));          // amc - i bmd
    yf[q + s * (4 * p + 2)] = cmul(apc - bpd, tw_lookup(tw, 2 * i1));
    yf[q + s * (4 * p + 3)] = cmul(sub_mi(amc, bmd), tw_lookup(tw, 3 * i1));      // amc + i bmd
}

__global__ __launch_bounds__(kThreads) void gen_mul_conj_kernel(cf* __restrict__ a, const cf* __restrict__ bhat, int M,
                                                               long total)
{
    const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
    if (i >= total) return;
    cf z = cmul(a[i], bhat[i % M]);
    z.y = -z.y;
    a[i] = z;
}

// pwr[k] (+)= sum_f |X[f][k]|^2 for k < N: squares and sums in double (datastore.cxx:83-85)
__global__ __launch_bounds__(kThreads) void gen_accum_kernel(const cf* __restrict__ X, int N, int M, int frames,
                                                            double* __restrict__ pwr, int accumulate)
{
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= N) return;
    double s = accumulate ? pwr[k] : 0.0;
    for (int f = 0; f < frames; ++f) {
        const cf v = X[static_cast<size_t>(f) * M + k];
        const double re = static_cast<double>(v.x), im = static_cast<double>(v.y);
        s = __builtin_fma(im, im, __builtin_fma(re, re, s));
    }
    pwr[k] = s;
}

int ilog2i(long v)
{
    int l = 0;
    while ((1L << l) < v) ++l;
    return l;
}

}  // namespace

// N even, not covered by the tuned families; pow2 N <= 2^26, other even N <= 2^23 (Bluestein length M <= 2^24).
bool generic_supported(int N)
{
    if (N < 2 || (N & 1)) return false;
    const bool pow2 = (N & (N - 1)) == 0;
    return pow2 ? N <= (1 << 26) : N <= (1 << 23);
}

int generic_length(int N)       // transform length: N itself, or Bluestein's M
{
    if ((N & (N - 1)) == 0) return N;
    int M = 64;
    while (M < 2 * N - 1) M *= 2;
    return M;
}

int generic_batch(int N)        // frames per batch: about 2^22 complex values in flight
{
    const long M = generic_length(N);
    return static_cast<int>(std::max<long>(1, std::min<long>(64, (1L << 22) / M)));
}

size_t generic_scratch_bytes(int N)     // two ping-pong buffers of batch x M complex
{
    return 2 * sizeof(cf) * static_cast<size_t>(generic_length(N)) * generic_batch(N);
}

// Host tables: t0 (2^h entries), t1 (M / 2^h entries) of W_M; *h_out = h.
void generic_twiddle_tables(int N, std::vector<cf>& t0, std::vector<cf>& t1, int* h_out)
{
    const long M = generic_length(N);
    const int logM = ilog2i(M), h = (logM + 1) / 2;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    t0.resize(static_cast<size_t>(1) << h);
    t1.resize(static_cast<size_t>(M >> h));
    for (size_t j = 0; j < t0.size(); ++j) {
        const long double a = two_pi * static_cast<long double>(j) / static_cast<long double>(M);
        t0[j] = cf{static_cast<float>(cosl(a)), static_cast<float>(-sinl(a))};
    }
    for (size_t j = 0; j < t1.size(); ++j) {
        const long double a = two_pi * static_cast<long double>(j << h) / static_cast<long double>(M);
        t1[j] = cf{static_cast<float>(cosl(a)), static_cast<float>(-sinl(a))};
    }
    *h_out = h;
}

// One Stockham transform of `frames` frames of length M in d_a (ping-pong with d_b);
// returns the buffer that holds the result.
static cf* run_fft(cf* d_a, cf* d_b, long M, int frames, const cf* d_t0, const cf* d_t1, int h, hipStream_t stream)
{
    const int logM = ilog2i(M);
    TwoLevel tw{d_t0, d_t1, h, (1u << h) - 1u};
    cf* src = d_a;
    cf* dst = d_b;
    int ls = 0;
    if (logM & 1) {
        const long total = static_cast<long>(frames) * (M / 2);
        hipLaunchKernelGGL(gen_radix2_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads),
                           0, stream, src, dst, static_cast<int>(M), ls, total, tw);
        std::swap(src, dst);
        ls = 1;
    }
    for (; ls < logM; ls += 2) {
        const long total = static_cast<long>(frames) * (M / 4);
        hipLaunchKernelGGL(gen_radix4_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads),
                           0, stream, src, dst, static_cast<int>(M), ls, total, tw);
        std::swap(src, dst);
    }
    return src;
}

// Frames [0, nframes) of d_stream (frame f = bytes [2N f, 2N (f+1))) -> d_pwr[N]
// (overwritten unless accumulate).  d_window: N floats or null (power-of-two N);
// d_g / d_bhat: bluestein_tables.h's tables (other N).  d_scratch: generic_scratch_bytes(N).
hipError_t launch_generic(int N, const uint8_t* d_stream, long nframes, const float* d_window, const cf* d_g,
                          const cf* d_bhat, const cf* d_t0, const cf* d_t1, int h, cf* d_scratch, double* d_pwr,
                          bool accumulate, hipStream_t stream)
{
    if (!generic_supported(N) || nframes < 1) return hipErrorInvalidValue;
    const long M = generic_length(N);
    const bool blu = M != N;
    const int batch = generic_batch(N);
    cf* const d_a = d_scratch;
    cf* const d_b = d_scratch + static_cast<size_t>(M) * batch;
    bool acc = accumulate;
    for (long done = 0; done < nframes; done += batch) {
        const int nb = static_cast<int>(std::min<long>(batch, nframes - done));
        const uint8_t* src = d_stream + static_cast<size_t>(done) * 2 * N;
        const long total = static_cast<long>(nb) * M;
        const unsigned blocks = static_cast<unsigned>((total + kThreads - 1) / kThreads);
        if (blu)
            hipLaunchKernelGGL(gen_load_kernel<true>, dim3(blocks), dim3(kThreads), 0, stream, src, N, static_cast<int>(M),
                               total, d_window, d_g, d_a);
        else
            hipLaunchKernelGGL(gen_load_kernel<false>, dim3(blocks), dim3(kThreads), 0, stream, src, N, static_cast<int>(M),
                               total, d_window, d_g, d_a);
        cf* res = run_fft(d_a, d_b, M, nb, d_t0, d_t1, h, stream);
        if (blu) {
            hipLaunchKernelGGL(gen_mul_conj_kernel, dim3(blocks), dim3(kThreads), 0, stream, res, d_bhat, static_cast<int>(M),
                               total);
            cf* other = (res == d_a) ? d_b : d_a;
            res = run_fft(res, other, M, nb, d_t0, d_t1, h, stream);
        }
        hipLaunchKernelGGL(gen_accum_kernel, dim3((N + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, res, N,
                           static_cast<int>(M), nb, d_pwr, acc ? 1 : 0);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        acc = true;
    }
    return hipSuccess;
}

}  // namespace rpf
