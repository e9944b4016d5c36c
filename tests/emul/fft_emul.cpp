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

}  // namespace

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
