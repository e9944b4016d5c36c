// rpf_fourstep.hip -- gfx950 kernels for transform lengths that do not fit one
// workgroup's LDS: N = N1*N2 with N1, N2 in {128, 256, 512}, i.e. the powers of
// two 16384 ... 262144 (config C4 of BASELINE.json is N = 262144 = 512 x 512).
//
// Four-step decomposition, n = N2 n1 + n2, k = k1 + N1 k2:
//
//   X[k1 + N1 k2] = sum_{n2} W_N2^{n2 k2} * ( W_N^{n2 k1} * sum_{n1} x[N2 n1 + n2] W_N1^{n1 k1} )
//
//   K2a fourstep_cols_kernel  for a tile of 64 columns n2: the raw u8 rows (128 B =
//        one cache line per n1) are staged in LDS by dword LDS-DMA into rows padded to
//        33 dwords (conflict-free column reads); every wavefront runs N1-point
//        column FFTs -- 8 points per lane, so N1/8 lanes per column and 512/N1
//        columns side by side in one wave -- with wave-local LDS exchanges (no
//        s_barrier), multiplies by W_N^{n2 k1} and writes Y[frame][n2][k1] as
//        coalesced rows.  (-1)^n = (-1)^n2 is a per-column constant.  The inter-step
//        twiddles and the window are read from tables the host lays out in the
//        kernel's own lane order (lane_ordered_twiddles / transposed_window below), so
//        that a wave's load is one contiguous run instead of a 64-line gather.
//   K2b fourstep_rows_kernel  workgroup (k1 tile of 16*(512/N2) rows, frame group):
//        loads the [N2 n2][tile k1] slab of Y (whole 128-byte lines), every
//        wavefront owns 512/N2 rows for the whole launch: N2-point FFT over n2,
//        |X|^2 into 8 per-lane f64 register accumulators; at the end the rows of a
//        tile leave as whole 128-byte lines of the per-frame-group partial spectrum.
//   K3  (rpf_kernels.hip) sums the frame-group partials into pwr.
//
// HBM/L2 traffic per sample: 2 B raw (algorithmic) + 8 B Y written + 8 B Y read
// + 8 B of W_N twiddles (L2-resident table); frames are processed in batches of
// 2 GB of Y -- 1024 frames at N = 262144, i.e. config C4's whole acquisition in ONE launch pair
// (round 3; 288 GB of HBM are there to be used).  Measured: 256 MB batches 1.19 ms per C4
// acquisition, 2 GB 1.03 ms; round 1: 64 MB batches 13 % slower than 256 MB, 128 MB 2 % -- launch
// tails and K2b's partial-spectrum round trip per launch decide, not Infinity-Cache residency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <cstdlib>
#include <thread>

#include "dft_small_wide.h"
#include "fused_layout.h"
#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

struct __attribute__((aligned(16))) cf4 { cf lo, hi; };   // two neighbouring complex values = one 16-byte access

constexpr int kWG = 1024, kWaves = kWG / 64;
constexpr int kColTile = 64;          // columns per K2a tile (128 raw bytes per row)
constexpr int kRowDwords = 33;        // 32 data dwords + 1 pad per staged raw row
constexpr size_t kScratchBytes = static_cast<size_t>(2048) << 20;      // of intermediate per launch pair
// LDS set aside behind the rows kernels' slabs for the exact twiddles of the pass before the last (fourstep_wide2 below):
// up to 64 entries of 16 bytes + room to align the table to 16 bytes
constexpr int kWideTabBytes = 64 * 16 + 16;

// Everything that depends on the factorisation N = N1 * N2.
template <int N1_, int N2_>
struct Split {
    static constexpr int N1 = N1_, N2 = N2_, N = N1_ * N2_;
    using GA = Geom<N1, 8>;                       // column transform (over n1)
    using GB = Geom<N2, 8>;                       // row transform (over n2)
    static constexpr int SUBA = 64 / GA::T;       // columns one wave transforms side by side
    static constexpr int SUBB = 64 / GB::T;       // rows one wave transforms side by side
    static constexpr int COLS_PER_WAVE = kColTile / kWaves;             // 4
    static constexpr int ROW_TILE = 16 * SUBB;    // k1 rows per K2b workgroup
    static constexpr int ROW_PITCH = ROW_TILE + 1;
    static constexpr int ROW_TILES = N1 / ROW_TILE;
    static constexpr int SLAB_A = SUBA * GA::LDS_CPX;                   // complex per wave
    static constexpr int SLAB_B = SUBB * GB::LDS_CPX;
    static constexpr int COLS_LDS = N1 * kRowDwords * 4 + kWaves * SLAB_A * (int)sizeof(cf);
    static constexpr int ROWS_LDS = N2 * ROW_PITCH * (int)sizeof(cf) + kWaves * SLAB_B * (int)sizeof(cf) + kWideTabBytes;
    static constexpr int BATCH = (int)(kScratchBytes / (sizeof(cf) * (size_t)N));   // frames per launch pair
    static constexpr int GROUPS = (256 / ROW_TILES) < BATCH ? (256 / ROW_TILES) : BATCH;   // frame groups
    static_assert(GA::T <= 64 && GB::T <= 64 && COLS_PER_WAVE % SUBA == 0, "");
    static_assert(COLS_LDS <= 160 * 1024 && ROWS_LDS <= 160 * 1024, "");
};

// N-point FFT (G = Geom<N, 8>) of the 8 values per lane of a T-lane group (pass-1
// layout: lane t holds elements t + T a); leaves X[bin_of<G>(t, a)] in register a.
template <class G>
__device__ __forceinline__ void group_fft(int t, cf* x, const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab)
{
    middle_passes<G, 1>(t, x, tw, slab);
    phase_fetch<G, G::NPASS>(t, x, slab);
    phase_last<G>(x);
}
// The row transform's LAST pass, and pwr += |X|^2 from it.  x: the values as fetched for the last pass; acc[a] is register
// a's bin as after phase_last.  WIDE: that pass in double, like the split forms' (dft_small_wide.h) -- the pass in which a
// line's energy has collected in one butterfly, so that every float32 rounding inside it lands on the weak bins beside
// the line.  Shipped from kFourstepWideFrom bins up (round 6): measured (profiles/r05_fourstep_wide.txt,
// r06_fourstep_wide.txt) the GPU then sits 0.4 - 1.4e-6 from float64 truth at 131072 / 262144 bins instead of 1.3 - 2.5e-6,
// for 2 % of C4's rate (and 2.1 - 7.1e-7 with the exact twiddles below).  It also moves the GPU AWAY from oracle/rpf_oracle.c on the line bins (C4: 4.8e-7 -> 1.6e-6): at these
// power-of-two lengths the oracle's radix-4 / 2 float32 butterflies and the float32 pass make the same roundings beside a line
// and agree with each other far better than either agrees with the truth.  The reference's arithmetic is FFTW's
// (/root/reference/src/datastore.cxx:82), whose codelets share neither's roundings, and |gpu - FFTW| <= |gpu - truth| +
// |FFTW - truth|: the distance from the truth is the objective, the agreement with the (unpinned) oracle was an artefact.
// -DRPF_FOURSTEP_WIDE=0 (`make f32pass`, A/B only) builds every size on the float32 pass, =1 every size on the wide one.
#ifndef RPF_FOURSTEP_WIDE
#define RPF_FOURSTEP_WIDE -1
#endif
constexpr int kFourstepWideFrom = 65536;
constexpr bool fourstep_is_wide(int n) { return RPF_FOURSTEP_WIDE < 0 ? n >= kFourstepWideFrom : RPF_FOURSTEP_WIDE != 0; }
template <class G, bool WIDE>
__device__ __forceinline__ void last_pass_accumulate(cf* x, double* acc)
{
    if constexpr (WIDE) {
        constexpr int R = G::RLAST;
#pragma unroll
        for (int g = 0; g < G::P / R; ++g) {
            cd w[R];
#pragma unroll
            for (int n = 0; n < R; ++n) w[n] = cd{static_cast<double>(x[g * R + n].x), static_cast<double>(x[g * R + n].y)};
            WideDft<R>::run(w);
#pragma unroll
            for (int k = 0; k < R; ++k) acc[g * R + k] = __builtin_fma(w[k].y, w[k].y, __builtin_fma(w[k].x, w[k].x, acc[g * R + k]));
        }
    } else {
        phase_last<G>(x);
        phase_accumulate(x, acc, G::P);
    }
}
// The pass BEFORE the last multiplies by twiddles that are EXACT to double precision, not the float32 table's values.
// Measured (profiles/r06_fourstep_wide.txt): with only the last pass in double 131072 bins stayed 0.4 - 1.4e-6 from
// float64 truth, its worst bin always a weak deterministic LINE (bin 2N/16) that shares its last two butterflies with a
// strong one; a float32 emulation of the transform (tools/analysis/fourstep_passes.py) keeps 1.3e-6 there with EVERY pass
// in double as long as the twiddle TABLE is float32: the table's representation error (1.7e-8 on 1/sqrt(2)) is a
// deterministic leak from the strong line into the weak one, the same in every frame, so it never averages down.  The
// lines' path through the earlier passes (and through the whole column transform: lines sit at k1 = 0) only meets the
// twiddle 1.  Three forms, RPF_FOURSTEP_WIDE2 =
//   1 (shipped)  the pass's butterflies stay float32; each product is x (hi + lo), hi the float32 table's value and
//                lo = float(exact - hi), read as one 16-byte (hi, lo) pair from a 0.4 - 0.9 KB LDS table: two more packed
//                instructions per product, 14 per thread and transform, no more LDS instructions than before.
//                Emulated 2.3 - 5.3e-7 from the truth at 65536 ... 262144 bins; measured 2.1 - 7.1e-7 at 131072 / 262144
//                on six streams x two windows, - 2.1 % on C4 beyond the double last pass.
//   2 (A/B)      the whole pass in double with double twiddles: emulated 2.0 - 3.0e-7, measured 1.9 - 6.8e-7, - 5.5 % on C4.
//   0 (A/B)      the last pass alone in double: up to 1.42e-6 at 131072 bins.
#ifndef RPF_FOURSTEP_WIDE2
#define RPF_FOURSTEP_WIDE2 1
#endif
// From kFourstepExactFrom bins up: at 65536 the last pass in double alone holds 1e-6 against the CPU path AND the truth on
// every recorded stream (<= 7.8e-7 / 7.6e-7), and form 1 missed the CPU path by a hair on one (1.004e-6, the CPU path
// itself 7.2e-7 from the truth there) -- the contract's comparator decides where it can.
constexpr int kFourstepExactFrom = 131072;
template <class G>
constexpr int fourstep_wide2(int n) { return fourstep_is_wide(n) && n >= kFourstepExactFrom && G::NPASS >= 3 ? RPF_FOURSTEP_WIDE2 : 0; }
// W_L^k = (cos, -sin)(2 pi k / L), k < L, in double (constexpr Taylor series of dft_small.h, ~1e-16)
template <int L>
struct WideTwiddleTable {
    double c[L], s[L];
    constexpr WideTwiddleTable() : c(), s()
    {
        for (int k = 0; k < L; ++k) {
            c[k] = cos_turn(k, L);
            s[k] = -sin_turn(k, L);
        }
    }
};
template <int L>
__device__ constexpr WideTwiddleTable<L> kWideTwiddles{};
// The kernels read them from LDS (filled once per launch; from global memory per transform they cost the fused kernel
// 28 %: scattered loads in a role whose memory queue is the tile loads'): form 2 as 16-byte doubles indexed by the
// exponent; form 1 as (hi, lo) float pairs, one row of P - 1 per m = t mod L_J, laid out like fill_twlds lays out the
// table's own values.
static_assert(kWideTabBytes == 64 * (int)sizeof(cd) + 16, "L <= 64 entries + room to align to 16 bytes");
template <class G>
constexpr bool fourstep_has_wide_table(int n) { return fourstep_wide2<G>(n) != 0; }
// (the offset from the start of the LDS allocation is rounded up, not the pointer's integer value: a pointer that has been
// through an integer loses its address space and every read of the table becomes a FLAT load on the vector memory path --
// measured: that alone cost C4 16 %)
template <class T>
__device__ __forceinline__ cd* wide_table_at(unsigned char* smem, T* after)
{
    const int off = static_cast<int>(reinterpret_cast<unsigned char*>(after) - smem);
    return reinterpret_cast<cd*>(smem + ((off + 15) & ~15));
}
// twN: the master table W_N^k the pass's float32 twiddles come from (load_twiddles / fill_twlds read the same entries)
template <class G, int MODE>
__device__ __forceinline__ void fill_wide_table(cd* tab, int tid, int nthreads, const cf* __restrict__ twN)
{
    constexpr int J = G::NPASS - 1, L = G::Lprev(J);
    static_assert(L <= 64 && G::Lcur(J) * (G::P - 1) * (int)sizeof(cf4) <= 64 * (int)sizeof(cd), "kWideTabBytes");
    if constexpr (MODE == 2) {
        for (int k = tid; k < L; k += nthreads) tab[k] = cd{kWideTwiddles<L>.c[k], kWideTwiddles<L>.s[k]};
    } else {
        cf4* const hl = reinterpret_cast<cf4*>(tab);             // {hi, lo} side by side: one 16-byte LDS read per product
        for (int i = tid; i < G::Lcur(J) * (G::P - 1); i += nthreads) {
            const int m = i / (G::P - 1), r = i % (G::P - 1) + 1;
            const cf hi = twN[m * r * ipow(G::P, J - 1)];
            cf4 e;
            e.lo = hi;
            e.hi = cf{static_cast<float>(kWideTwiddles<L>.c[m * r] - static_cast<double>(hi.x)),
                      static_cast<float>(kWideTwiddles<L>.s[m * r] - static_cast<double>(hi.y))};
            hl[i] = e;
        }
    }
}
// a (hi + lo): the small product first, the two large terms folded onto it -- four packed instructions
__device__ __forceinline__ cf cmul_compensated(cf a, cf hi, cf lo)
{
    cf d;
    asm("v_pk_mul_f32 %0, %1, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(d)
        : "v"(a), "v"(hi), "v"(lo));
    return d;
}
// passes J ... NPASS-1 of a row transform (middle_passes of rpf_device_common.h), the pass before the last in form MODE
template <class G, int J, bool TWLDS, int MODE>
__device__ __forceinline__ void row_passes(int t, cf* x, const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab, const cf* twtable,
                                           const cd* widetab)
{
    if constexpr (J < G::NPASS) {
        if constexpr (J > 1) phase_fetch<G, J>(t, x, slab);
        cf twj[G::P - 1];
        if constexpr (MODE == 1 && J == G::NPASS - 1) {
            // (hi, lo) pairs from the wide table below, not from the registers / the twiddle table
        } else if constexpr (TWLDS && J >= 2) {
            const cf* row = twtable + twlds_offset<G, J>() + (t % G::Lcur(J)) * (G::P - 1);
#pragma unroll
            for (int r = 0; r < G::P - 1; ++r) twj[r] = row[r];
        } else {
#pragma unroll
            for (int r = 0; r < G::P - 1; ++r) twj[r] = tw[J - 1][r];
        }
        if constexpr (MODE == 2 && J == G::NPASS - 1) {
            cd w[G::P];
#pragma unroll
            for (int a = 0; a < G::P; ++a) w[a] = cd{static_cast<double>(x[a].x), static_cast<double>(x[a].y)};
            WideDft<G::P>::run(w);
            x[0] = cf{static_cast<float>(w[0].x), static_cast<float>(w[0].y)};
            const int m = t % G::Lcur(J);                // W_{L_{J-1}}^{m r}, m = t mod L_J (fft_core.h twiddle_index)
#pragma unroll
            for (int r = 1; r < G::P; ++r) {
                const cd wr = widetab[m * r];
                const cd v = wide_cmul(w[r], wr.x, wr.y);
                x[r] = cf{static_cast<float>(v.x), static_cast<float>(v.y)};
            }
        } else if constexpr (MODE == 1 && J == G::NPASS - 1) {
            const cf4* hl = reinterpret_cast<const cf4*>(widetab) + (t % G::Lcur(J)) * (G::P - 1);
            Dft<G::P>::run(x);
#pragma unroll
            for (int r = 1; r < G::P; ++r) {
                const cf4 e = hl[r - 1];
                x[r] = cmul_compensated(x[r], e.lo, e.hi);
            }
        } else {
            phase_butterfly_twiddle<G>(x, twj);
        }
        phase_store<G, J>(t, x, slab);
        exchange_sync<(G::Lcur(J) > 64)>();
        row_passes<G, J + 1, TWLDS, MODE>(t, x, tw, slab, twtable, widetab);
    }
}
// group_fft without its last pass's butterflies: the values as phase_fetch leaves them
template <class G, bool TWLDS, int MODE = 0>
__device__ __forceinline__ void group_fft_but_last(int t, cf* x, const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab, const cf* twtable,
                                                   const cd* widetab = nullptr)
{
    row_passes<G, 1, TWLDS, MODE>(t, x, tw, slab, twtable, widetab);
    phase_fetch<G, G::NPASS>(t, x, slab);
}
// The same with the twiddles of the passes >= 2 read from an LDS table (fill_twlds) instead of held in registers:
// 2 (P - 1) VGPRs fewer per such pass, which the fused kernel's roles need for their per-lane constants.
template <class G>
__device__ __forceinline__ void group_fft_twlds(int t, cf* x, const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab, const cf* twtable)
{
    PhaseClock none;
    middle_passes<G, 1, 0, true>(t, x, tw, slab, none, twtable);
    phase_fetch<G, G::NPASS>(t, x, slab);
    phase_last<G>(x);
}

// BLU: first step of the large Bluestein path (see bluestein_mid_kernel): the
// frame has n_true < S::N samples, a[n] = (v[n] - 127) g[n] zero-padded to S::N
// (g carries (-1)^n, the window and the chirp), frames are n_true samples apart.
template <class S, bool WINDOW, bool DMA, bool BLU = false>
__global__ __launch_bounds__(kWG, 4) void fourstep_cols_kernel(const uint8_t* __restrict__ stream,
                                                              int nframes,
                                                              const cf* __restrict__ tw_sub,
                                                              const cf* __restrict__ twN,
                                                              const float* __restrict__ window,
                                                              cf* __restrict__ Y, int n_true,
                                                              const cf* __restrict__ g)
{
    using G = typename S::GA;
    constexpr int N1 = S::N1, N2 = S::N2, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* const raw = smem;                                                 // [N1][33] dwords
    cf* const slabs = reinterpret_cast<cf*>(smem + N1 * kRowDwords * 4);       // [16][SLAB_A]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int sub = lane / T, t = lane % T;
    cf* const slab = slabs + wave * S::SLAB_A + sub * G::LDS_CPX;

    cf tw[G::NPASS - 1][G::P - 1];
    load_twiddles<G, 1>(t, tw_sub, tw);

    constexpr int TILES = N2 / kColTile;
    const int ntasks = nframes * TILES;
#pragma unroll 1
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int f = task / TILES, ct = task % TILES;
        const uint8_t* const frame = stream + static_cast<size_t>(f) * (2 * (BLU ? n_true : S::N));

        __syncthreads();   // the previous tile has been consumed by every wave
        // stage the [N1 rows][128 B] raw tile: LDS dword L <- row L/33, dword L%33
#pragma unroll 1
        for (int i = 0; i < (N1 * kRowDwords + kWG - 1) / kWG; ++i) {
            const int L = i * kWG + tid;
            const int r = L / kRowDwords, d = L % kRowDwords;
            // BLU: sample pairs past the end of the frame are never read (they are zeros)
            if (L < N1 * kRowDwords && d < 32 && (!BLU || N2 * r + kColTile * ct + 2 * d < n_true)) {
                const uint8_t* src = frame + 2 * (static_cast<size_t>(N2) * r + kColTile * ct) + 4 * d;
                if constexpr (DMA) {
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(raw + 4 * (i * kWG + wave * 64)),
                                                     4, 0, 0);
                } else {
                    const uint16_t lo = *reinterpret_cast<const uint16_t*>(src);
                    const uint16_t hi = *reinterpret_cast<const uint16_t*>(src + 2);
                    *reinterpret_cast<uint32_t*>(raw + 4 * L) = lo | (static_cast<uint32_t>(hi) << 16);
                }
            }
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

#pragma unroll 1
        for (int j = 0; j < S::COLS_PER_WAVE / S::SUBA; ++j) {
            const int cl = S::COLS_PER_WAVE * wave + S::SUBA * j + sub;   // column inside the tile
            const int c = kColTile * ct + cl;                             // n2
            const float sgn = (c & 1) ? -1.0f : 1.0f;                     // (-1)^n, n = N2 n1 + n2
            const float off = -(kTwo23 + 127.0f) * sgn;
            cf x[G::P];
#pragma unroll
            for (int a = 0; a < G::P; ++a) {
                const int n1 = t + T * a;
                const uint32_t iq =
                    *reinterpret_cast<const uint16_t*>(raw + 4 * (n1 * kRowDwords + (cl >> 1)) + 2 * (cl & 1));
                const cf v = iq_plus_2p23(iq);
                if constexpr (BLU) {
                    // g is zero past the end of the frame (whatever the unread LDS bytes hold is finite)
                    x[a] = cmul(v - (kTwo23 + 127.0f), g[static_cast<size_t>(c) * N1 + n1]);
                } else if constexpr (WINDOW) {
                    const float w = window[static_cast<size_t>(c) * N1 + n1] * sgn;
                    x[a] = (v - (kTwo23 + 127.0f)) * w;
                } else {
                    x[a] = v * sgn + off;
                }
            }
            group_fft<G>(t, x, tw, slab);
            // inter-step twiddle W_N^{n2 k1} (n2 k1 < N: no reduction needed), then
            // through the group's slab into natural k1 order for a coalesced row store
            exchange_sync<false>();
#pragma unroll
            for (int a = 0; a < G::P; ++a) {
                const int k1 = bin_of<G>(t, a);
                slab[G::slot(k1)] = cmul(x[a], twN[static_cast<size_t>(c) * N1 + T * a + t]);
            }
            exchange_sync<false>();
            // 16 bytes per lane and store: 8-byte stores are issue-bound at ~7 B/clk/CU
            // (MI355X_MICROARCH.md, store tail), which is what K2a ran at
            // Layout of the intermediate.  Four-step: tile-major, Y[f][k1 / ROW_TILE][n2][k1 % ROW_TILE] -- the
            // [N2][ROW_TILE] tile K2b loads is then ONE contiguous run (64 KB at 512 x 512) instead of N2 lines
            // 8 N1 bytes apart, and what a K2a workgroup adds to a tile is kColTile consecutive lines.  The large
            // Bluestein path keeps Y[f][n2][k1] (its consumer, bluestein_mid_kernel, walks whole rows).
            cf* const yrow = BLU ? Y + (static_cast<size_t>(f) * N2 + c) * N1
                                 : Y + static_cast<size_t>(f) * S::N + static_cast<size_t>(c) * S::ROW_TILE;
#pragma unroll
            for (int a = 0; a < G::P / 2; ++a) {
                const int e = 2 * t + 2 * T * a;
                cf4 v;
                v.lo = slab[G::slot(e)];
                v.hi = slab[G::slot(e + 1)];
                if constexpr (BLU) *reinterpret_cast<cf4*>(yrow + e) = v;
                else *reinterpret_cast<cf4*>(yrow + static_cast<size_t>(e / S::ROW_TILE) * (N2 * S::ROW_TILE) + e % S::ROW_TILE) = v;
            }
            exchange_sync<false>();
        }
    }
}

// TILED: the intermediate is tile-major (K2a's four-step layout above); else Y[f][n2][k1] (large Bluestein).
template <class S, bool TILED>
__global__ __launch_bounds__(kWG, 4) void fourstep_rows_kernel(const cf* __restrict__ Y, int nframes,
                                                              const cf* __restrict__ tw_sub,
                                                              double* __restrict__ partial, int first)
{
    using G = typename S::GB;
    constexpr int N1 = S::N1, N2 = S::N2, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const tile = reinterpret_cast<cf*>(smem);                                   // [N2][ROW_PITCH]
    cf* const slabs = tile + N2 * S::ROW_PITCH;                                     // [16][SLAB_B]
    cd* const widetab = wide_table_at(smem, slabs + kWaves * S::SLAB_B);                  // exact twiddles of the pass before the last

    const int tid = threadIdx.x;
    if constexpr (fourstep_has_wide_table<G>(S::N))                                  // (the loop's first barrier covers it)
        fill_wide_table<G, fourstep_wide2<G>(S::N)>(widetab, tid, kWG, tw_sub);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int sub = lane / T, t = lane % T;
    const int jrow = wave * S::SUBB + sub;                 // this lane group's row inside the tile
    cf* const slab = slabs + wave * S::SLAB_B + sub * G::LDS_CPX;
    const int ktile = blockIdx.x % S::ROW_TILES, fg = blockIdx.x / S::ROW_TILES;
    const int ngroups = gridDim.x / S::ROW_TILES;

    cf tw[G::NPASS - 1][G::P - 1];
    load_twiddles<G, 1>(t, tw_sub, tw);
    double acc[G::P];
#pragma unroll
    for (int a = 0; a < G::P; ++a) acc[a] = 0.0;

    // The next frame's tile is fetched into registers while this one is transformed
    // (one 1024-thread workgroup per CU: nothing else would overlap the two; a second
    // tile in flight needs 16 more VGPRs than the 128 available -- measured: spills, slower).
    constexpr int PER = N2 * S::ROW_TILE / kWG / 2;      // 16-byte loads: two neighbouring k1 per lane
    constexpr int HALF = S::ROW_TILE / 2;
    cf4 nxt[PER];
    auto fetch = [&](int f) {
        const cf* const yf = Y + static_cast<size_t>(f) * S::N + (TILED ? static_cast<size_t>(N2) * S::ROW_TILE : S::ROW_TILE) * ktile;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * kWG + tid;
            if constexpr (TILED) nxt[i] = *reinterpret_cast<const cf4*>(yf + 2 * idx);
            else nxt[i] = *reinterpret_cast<const cf4*>(yf + static_cast<size_t>(idx / HALF) * N1 + 2 * (idx % HALF));
        }
    };
    if (fg < nframes) fetch(fg);
#pragma unroll 1
    for (int f = fg; f < nframes; f += ngroups) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * kWG + tid;
            cf* const dst = tile + (idx / HALF) * S::ROW_PITCH + 2 * (idx % HALF);
            dst[0] = nxt[i].lo;
            dst[1] = nxt[i].hi;
        }
        __syncthreads();
        if (f + ngroups < nframes) fetch(f + ngroups);
        cf x[G::P];
#pragma unroll
        for (int a = 0; a < G::P; ++a) x[a] = tile[(t + T * a) * S::ROW_PITCH + jrow];
        group_fft_but_last<G, false, fourstep_wide2<G>(S::N)>(t, x, tw, slab, nullptr, widetab);
        last_pass_accumulate<G, fourstep_is_wide(S::N)>(x, acc);
        exchange_sync<false>();
    }

    // bins k = k1 + N1 k2: for a fixed k2 the tile's rows are ROW_TILE consecutive
    // doubles = whole 128-byte lines of the partial spectrum
    __syncthreads();
    double* const stage = reinterpret_cast<double*>(smem);                          // [N2 k2][ROW_PITCH]
#pragma unroll
    for (int a = 0; a < G::P; ++a) stage[bin_of<G>(t, a) * S::ROW_PITCH + jrow] = acc[a];
    __syncthreads();
    double* const out = partial + static_cast<size_t>(fg) * S::N + S::ROW_TILE * ktile;
#pragma unroll
    for (int i = 0; i < N2 * S::ROW_TILE / kWG; ++i) {
        const int idx = i * kWG + tid;
        const int k2 = idx / S::ROW_TILE, j = idx % S::ROW_TILE;
        double* p = out + static_cast<size_t>(k2) * N1 + j;
        const double v = stage[k2 * S::ROW_PITCH + j];
        *p = first ? v : (*p + v);
    }
}


// ---- fused four-step: the intermediate is read back from the writing XCD's L2 --------------------
// K2a/K2b above exchange Y through the fabric: 8 B written + 8 B read per sample against 2 algorithmic bytes,
// and the pair runs at the speed of that traffic.  What the L2 offers (profiles/r04_l2_residency.txt):
// it is a WRITE-BACK cache -- up to ~2 MB of dirty lines per XCD stay in it, a larger footprint streams its stores
// through -- and a line stored by one CU is served to another CU OF THE SAME XCD from the L2 (sc1 loads): the read
// half of the round trip can stay on chip, and with one 2 MB buffer of Y per team part of the write half too (the
// shipped two-buffer form cycles 4 MB through a 4 MB L2: all of Y is written back once and about two thirds of
// its reads miss, see NBUF below).  Here ONE persistent launch does both
// steps: the 32 workgroups that share an XCD (one per CU, found by HW_REG_XCC_ID: correctness never rests on a
// block -> XCD guess) form a team that owns one round of FR = 262144 / N frames at a time, 2 MB of Y, double
// buffered in the team's 4 MB L2.  Inside a workgroup the sixteen wavefronts split into two ROLES that run their
// own loops and never meet at an s_barrier (role barriers are LDS counters), so that the memory phases of one
// role hide behind the arithmetic of the other:
//     producers (waves 0-7)   raw rows of the workgroup's 16 * SUBA columns -> LDS (dword LDS-DMA, a round ahead);
//                             two column groups per wave: N1-point transforms, times W_N^{n2 k1} (the columns of a
//                             workgroup never change: the step twiddles and the window are per-lane REGISTERS,
//                             loaded once -- no table traffic beside Y in the L2), through the wave's slab into
//                             natural k1 order, plain 16-byte stores into buffer (round & 1) once the team has
//                             finished reading round - 2 out of it; drain; `produced` += 1.
//     consumers (waves 8-15)  wait for the team's 32 arrivals on `produced`; the [N2][16 * SUBB] tile of the
//                             round comes in with sc1 loads (served by the L2, never by this CU's L1, which other
//                             CUs' stores do not refresh); `consumed` += 1; two row groups per wave: N2-point
//                             transforms, |X|^2 into f64 register accumulators that live for the whole launch.
// The counters are touched only by atomics that execute in the XCD's L2.  Every spin is bounded; a team that does
// not assemble (a CU busy with someone else's kernel, an unexpected XCD population) raises ctl->abort, everybody
// leaves, K3 is told to skip (its d_skip word is ctl->abort) and K3's companion kernel reports the launch to the
// host (and NaN-fills a caller-owned spectrum) -- the engine then runs the same bytes through K2a/K2b and keeps
// to them (rpf_engine.cpp, recover_fused): loud, never wrong, and never a lost acquisition on the queue path.
// The engine proves the path once at creation and otherwise keeps K2a/K2b.
struct FusedCtl {
    unsigned arrivals[8][32];     // [xcd][0]: members registered (own 128-byte line each)
    // One pair of counters per buffer, cumulative over the rounds that use the buffer: a producer arrives for round j only
    // after all 32 consumers of round j - NBUF, a consumer for round j only after all 32 producers of round j -- no
    // arrival of a later round of the SAME buffer can stand in for a missing one of an earlier round.  (One counter for
    // two buffers could: 31 arrivals of round j + 1 of round j + 1 = 32.)
    unsigned produced[8][4][32];  // [xcd][buffer][0]: producer arrivals, 32 per round (three buffers in the half-frame form)
    unsigned consumed[8][4][32];  // [xcd][buffer][0]: consumer arrivals, 32 per round (16 per half round, half-frame form)
    unsigned registered[32];      // [0]: workgroups registered, grid-wide
    unsigned abort[32];           // [0]: != 0 -> results invalid
};
#ifdef RPF_FUSED_PROFILE
// measurement knobs (rpf_debug_fused_knobs): [0] poll pause 0: s_sleep 2, 1: none, 2: s_sleep 8, 4: vector-path polls;
// [1] 1: tiles rotated by half a frame against the team ranks, 2: the tiles' places in Y rotated, 3: no raw rows (garbage
// in), 5 / 6: the raw rows' LDS-DMA issued after the column transforms; [2] consumers skip their transforms; [3] producers
// skip theirs (results are garbage with 2, 3)
__device__ int g_fused_knob[4];
#define FKNOB(i) g_fused_knob[i]
#else
#ifndef RPF_FUSED_KNOB1
#define RPF_FUSED_KNOB1 0
#endif
#define FKNOB(i) ((i) == 1 ? RPF_FUSED_KNOB1 : 0)      // (-DRPF_FUSED_KNOB1=n: knob 1 at compile time; round 4's A/B, profiles/r04_c4_fused.txt)
#endif
constexpr unsigned kSpinLimit = 4u << 20;      // global polls of ~0.1-3 us: >= 0.5 s
constexpr unsigned kLdsSpinLimit = 1u << 26;   // LDS polls of ~50 ns

__device__ __forceinline__ unsigned ctl_load(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Current value of a team counter: an sc1 load, written in assembly, is served by the XCD's L2 -- where the team's
// atomics execute -- and never by this CU's L1 (a plain or sc0 load is, stale for ever once the line is resident).
// Round 2 polled with a returning atomic add of zero: 64 pollers per XCD saturate the one L2 channel that owns the
// counter's line (~88 atomics per microsecond), and every stream that stripes over the channels -- the Y stores'
// drain, the tile loads -- then runs at that channel's pace (measured: 94 us per round instead of < 10).
__device__ __forceinline__ unsigned l2_read(const unsigned* p)
{
    unsigned r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// What the two roles of a workgroup share in LDS (static: the dynamic area is full).
struct FusedSync {
    unsigned bar[3];          // [0], [1]: role barriers, monotonically increasing arrival counts (8 per barrier);
                              // [2]: consumer waves that have taken their columns out of the tile (8 per round)
    unsigned seen[2][4];      // [0][buffer]: `consumed` as last polled by the producers' wave 0; [1][buffer]: `produced`, consumers'
    unsigned abort;           // a bounded spin ran out somewhere in this workgroup (or the grid's flag was seen)
    int team[3];              // xcd, rank, ok
};

// The same word through the SCALAR memory path (s_load_dword glc: past the scalar cache, to the L2): a queue of its own,
// not behind the CU's vector-memory traffic -- where a poll otherwise waits out the CU's own tile loads, Y stores and
// LDS-DMA (measured in the profile build: + 3 % with two buffers of Y per team, + 16 % with one).  What the shipped
// kernel polls with; the vector form stays for measurement (knob 0 = 4).
__device__ __forceinline__ unsigned l2_read_scalar(const unsigned* p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(a >> 32));
    const unsigned long long ua = (static_cast<unsigned long long>(hi) << 32) | lo;
    unsigned r;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(ua) : "memory");
    return r;
}

// s_waitcnt vmcnt(0), as the BUILTIN: the compiler's own wait-count bookkeeping sees it.  (Written in assembly it is
// invisible: the compiler carries the round's sixteen Y stores as still pending into the next round and puts
// s_waitcnt vmcnt(1) / vmcnt(0) in front of the first writes to their data registers -- which, in the hardware's count,
// are the two LDS-DMA instructions just issued in assembly.)  gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4,
// lgkmcnt 11:8.
__device__ __forceinline__ void drain_vector_memory()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

// A word of LDS, read in assembly: no compiler-placed s_waitcnt vmcnt(0) in front of it (an atomic load of LDS got one
// while the builtin form of the LDS-DMA was in flight; none of these words is ever a DMA target).  The low half of a
// generic pointer into LDS is the LDS address.
__device__ __forceinline__ unsigned lds_peek(const unsigned* p)
{
    unsigned r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(r)
                 : "v"(static_cast<unsigned>(reinterpret_cast<unsigned long long>(p)))
                 : "memory");
    return r;
}

// Barrier among the 8 waves of one role (no s_barrier: that one spans both roles).  `target` = 8 x the number of
// this role's barriers so far.  LDS operations of one wave execute in order, so the arrival follows its writes.
// (split form: role_arrive where the wave is done, role_wait where it needs the others -- nothing stalls in between)
__device__ __forceinline__ void role_arrive(FusedSync* sy, int role, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&sy->bar[role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool role_wait(FusedSync* sy, int role, unsigned target)
{
    unsigned spins = 0;
    while (lds_peek(&sy->bar[role]) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kLdsSpinLimit || lds_peek(&sy->abort)) return false;
    }
    return true;
}
__device__ __forceinline__ bool role_barrier(FusedSync* sy, int role, unsigned target, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&sy->bar[role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    while (lds_peek(&sy->bar[role]) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kLdsSpinLimit || lds_peek(&sy->abort)) return false;
    }
    return true;
}

// Wait until the team's counter reaches `target`: wave 0 of the role polls the L2 (one lane), everybody else
// watches the value it publishes in LDS.
// dma_pending (knob 1 = 6, measurement only): the wave has LDS-DMA in flight and keeps off the LDS until it has landed.
__device__ __forceinline__ bool team_wait(FusedCtl* ctl, FusedSync* sy, const unsigned* counter, unsigned* seen, unsigned target,
                                          bool poller, int lane, bool dma_pending = false)
{
    if (!dma_pending && lds_peek(seen) >= target) return true;
    unsigned spins = 0;
    if (poller && dma_pending) {
        for (;;) {
            const unsigned v = l2_read_scalar(counter);
            if (v >= target) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(seen, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return true;
            }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit || ((spins & 255u) == 0 && l2_read_scalar(&ctl->abort[0]) != 0)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    __hip_atomic_store(&ctl->abort[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return false;
            }
        }
    }
    if (dma_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (poller) {
        for (;;) {
            const unsigned v = FKNOB(0) == 4 ? __builtin_amdgcn_readfirstlane(l2_read(counter)) : l2_read_scalar(counter);
            if (v >= target) {
                if (lane == 0) __hip_atomic_store(seen, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return true;
            }
            if (FKNOB(0) == 0 || FKNOB(0) == 4) __builtin_amdgcn_s_sleep(2);
            else if (FKNOB(0) == 2) __builtin_amdgcn_s_sleep(8);
            if (++spins > kSpinLimit || lds_peek(&sy->abort) ||
                ((spins & 255u) == 0 && l2_read_scalar(&ctl->abort[0]) != 0)) {
                if (lane == 0) {
                    __hip_atomic_store(&ctl->abort[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return false;
            }
        }
    }
    while (lds_peek(seen) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kLdsSpinLimit || lds_peek(&sy->abort)) return false;
    }
    return true;
}

// -DRPF_FUSED_PROFILE: lane 0 of wave 0 of each role adds the wall-clock ticks (100 MHz) it spends in each segment
// of a round to g_fused_prof (printed by tools/gpu_fused_profile.py, which left the tree in round 6: `git show 71dca54:tools/gpu_fused_profile.py`); never in the shipped library.
#ifdef RPF_FUSED_PROFILE
__device__ unsigned long long g_fused_prof[16];
__device__ unsigned long long g_fused_prof_wg[256][16];     // the same per workgroup (32 xcd + rank), last launch
// absolute 100 MHz time stamps of the hand-offs, [workgroup = 32 xcd + rank][round < 64][event]: 0 producers arrived,
// 1 consumers saw `produced`, 2 consumers arrived, 3 producers saw `consumed` (that tool prints
// where the signals spend their time)
__device__ unsigned long long g_fused_trace[256][64][4];
#define FTRACE(ev) do { if (rw == 0 && lane == 0 && j < 64) g_fused_trace[32 * xcd + rank][j][ev] = wall_clock64(); } while (0)

struct FusedClock {
    unsigned long long last, sum[8];
    __device__ __forceinline__ void start() { for (int i = 0; i < 8; ++i) sum[i] = 0; last = wall_clock64(); }
    __device__ __forceinline__ void stamp(int i) { const unsigned long long now = wall_clock64(); sum[i] += now - last; last = now; }
    __device__ __forceinline__ void publish(int base, bool who, int wg = -1)
    {
        if (who) {
            for (int i = 0; i < 7; ++i) atomicAdd(&g_fused_prof[base + i], sum[i]);
            atomicAdd(&g_fused_prof[base + 7], 1ull);
            if (wg >= 0) for (int i = 0; i < 7; ++i) g_fused_prof_wg[wg][base + i] = sum[i];
        }
    }
};
#define FSTAMP(i) fclk.stamp(i)
#else
struct FusedClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void publish(int, bool, int = -1) {}
};
#define FSTAMP(i) ((void)0)
#define FTRACE(ev) ((void)0)
#endif

constexpr int kRoleWaves = kWaves / 2, kRoleThreads = kRoleWaves * 64;

template <class S>
constexpr int fused_lds_bytes()
{
    constexpr int slabs = kRoleWaves * (S::SLAB_A + S::SLAB_B) * (int)sizeof(cf);
    constexpr int raw = S::N1 * 16 * S::SUBA * 2;                        // [N1 rows][2 COLS bytes]: 16 KB
    constexpr int tile = S::N2 * S::ROW_PITCH * (int)sizeof(cf);
    constexpr int twtables = (twlds_entries<typename S::GA>() + twlds_entries<typename S::GB>()) * (int)sizeof(cf);
    constexpr int steptab = 16 * S::SUBA * 8 * (int)sizeof(cf);        // the register-index factor of W_N^{n2 k1}, per column
    return slabs + tile + raw + twtables + steptab + kWideTabBytes;
}

// NBUF: buffers of Y per team.  1: the round's 2 MB stay in the L2 for good (every read hits, 39 % of the writes never
// leave: 3.4 x the algorithmic traffic, measured) but every round waits out both hand-offs, 2 x 1.2 - 3.8 us on a busy
// CU: 0.22 Tsample/s; 2 (shipped): a round's hand-offs hide behind the other buffer's work, 4 MB of Y cycle through a
// 4 MB L2, all of it is written back once and about two thirds of the reads miss: 0.245.  NT (measurement only): the
// consumers' tile loads and the raw rows carry the non-temporal hint -- no effect either way (profiles/r04_c4_fused.txt).
// 3 (round 5's experiment, -DRPF_FUSED_NBUF=3; profiles/r05_c4_halfframe.txt): the HALF-FRAME form -- Y in three buffers of HALF a round (the k1 tiles
// [0, TPF/2) and [TPF/2, TPF) of every frame slot: 1 MB each), half round q = 2 j + h in buffer q mod 3; the producers
// signal each half as its stores have drained, the sixteen consumer workgroups of a half start while the other half is
// being stored; 3 MB of Y per 4 MB L2 (profiles/r04_l2_residency.txt: at that footprint the reads still hit).
#ifndef RPF_FUSED_NBUF
#define RPF_FUSED_NBUF 2
#endif
template <class S, bool WINDOW, bool DMA, int NBUF = RPF_FUSED_NBUF, int NT = 0>
__global__ __launch_bounds__(kWG, 4) void fourstep_fused_kernel(const uint8_t* __restrict__ stream, int nframes,
                                                               const cf* __restrict__ tw_n1,
                                                               const cf* __restrict__ tw_n2,
                                                               const cf* __restrict__ twN,
                                                               const float* __restrict__ window,
                                                               cf* __restrict__ Yall, double* __restrict__ partial,
                                                               FusedCtl* __restrict__ ctl)
{
    using GA = typename S::GA;
    using GB = typename S::GB;
    constexpr int N1 = S::N1, N2 = S::N2, N = S::N;
    constexpr int TA = GA::T, TB = GB::T, P = 8;
    constexpr int TPF = N / 8192;                   // tiles per frame (both steps)
    constexpr int FR = 32 / TPF;                    // frames per round
    constexpr int COLS = 16 * S::SUBA;              // columns per workgroup and round
    constexpr int ROWB = 2 * COLS;                  // bytes per staged raw row (32 ... 128), unpadded: 16-byte LDS-DMA pieces
    using Raw = RawStage<ROWB>;                     // where a row's bytes sit in LDS (fused_layout.h: piece-major at 64 / 128 bytes per row)
    constexpr int GROUPS = 2;                       // column / row groups per wave: 16 groups over 8 waves
    constexpr int RT = S::ROW_TILE, HALF = RT / 2;
    static_assert(TPF >= 1 && TPF <= 32 && N2 / COLS == TPF && N1 / RT == TPF, "tile counts");
    static_assert(fused_lds_bytes<S>() <= 160 * 1024 - 64, "LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const slabsA = reinterpret_cast<cf*>(smem);                                    // [8][SLAB_A]
    cf* const slabsB = slabsA + kRoleWaves * S::SLAB_A;                                // [8][SLAB_B]
    cf* const tile = slabsB + kRoleWaves * S::SLAB_B;                                  // [N2][ROW_PITCH]
    uint8_t* const raw = reinterpret_cast<uint8_t*>(tile + N2 * S::ROW_PITCH);         // [N1][ROWB] bytes
    cf* const twtabA = reinterpret_cast<cf*>(raw + N1 * ROWB);                    // later passes' twiddles, columns
    cf* const twtabB = twtabA + twlds_entries<GA>();                                   // ... and rows
    cf* const steptab = twtabB + twlds_entries<GB>();                                  // [COLS][P]: W_N^{n2 (bin_of(0, a))}
    cd* const widetab = wide_table_at(smem, steptab + COLS * P);                             // exact twiddles of the row transform's pass before the last
    __shared__ FusedSync sync_;
    FusedSync* const sy = &sync_;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    fill_twlds<GA, 1>(tid, kWG, tw_n1, twtabA);
    fill_twlds<GB, 1>(tid, kWG, tw_n2, twtabB);
    if constexpr (fourstep_has_wide_table<GB>(N)) fill_wide_table<GB, fourstep_wide2<GB>(N)>(widetab, tid, kWG, tw_n2);
    // ---- team assembly -------------------------------------------------------------------------
    if (tid == 0) {
        const int xcd = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20)) & 7;   // XCC_ID[3:0]
        const unsigned rank = __hip_atomic_fetch_add(&ctl->arrivals[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->registered[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (ctl_load(&ctl->registered[0]) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { ok = 0; break; }
        }
        // the static frame -> team map needs eight equal teams of 32
        for (int x = 0; ok && x < 8; ++x)
            if (ctl_load(&ctl->arrivals[x][0]) != 32u) ok = 0;
        if (!ok) __hip_atomic_store(&ctl->abort[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sy->team[0] = xcd;
        sy->team[1] = static_cast<int>(rank);
        sy->team[2] = ok;
        sy->bar[0] = sy->bar[1] = sy->bar[2] = 0;
        for (int b = 0; b < 4; ++b) sy->seen[0][b] = sy->seen[1][b] = 0;
        sy->abort = 0;
    }
    __syncthreads();                                 // the only workgroup-wide barrier before the output stage
    const int xcd = sy->team[0], rank = sy->team[1];
    if (!sy->team[2] || rank >= 32) return;
    // (measurement knob 1, profile build: the tiles rotated by half a frame against the ranks -- does a slow workgroup
    //  follow its tile or its CU?)
    const int fsl = rank / TPF, tl = (rank % TPF + (FKNOB(1) == 1 ? TPF / 2 : 0)) % TPF;            // frame slot of the round, tile of the frame
    const int YROT = FKNOB(1) == 2 ? 7 % TPF : 0;     // (knob 1 = 2: the tiles' places in Y rotated by 7 -- does the slow tile follow its address?)
    const int nrounds = (nframes + FR - 1) / FR;
    const int nj = xcd < nrounds ? (nrounds - xcd + 7) / 8 : 0;    // this team's rounds: xcd, xcd + 8, ...
    constexpr bool HALFF = NBUF == 3;                // half-frame form
    constexpr int HT = TPF / 2;                      // k1 tiles per half
    constexpr size_t HB = static_cast<size_t>(FR) * N / 2;      // complex values per half buffer
    static_assert(!HALFF || TPF % 2 == 0, "two halves of whole tiles");
    cf* const Yteam = HALFF ? Yall + static_cast<size_t>(xcd) * 3 * HB + static_cast<size_t>(fsl) * (N / 2)       // + (q % 3) * HB
                            : Yall + static_cast<size_t>(xcd) * NBUF * FR * N + static_cast<size_t>(fsl) * N;     // + (j % NBUF) * FR * N

    const bool producer = wave < kRoleWaves;
    const int rw = wave & (kRoleWaves - 1);          // wave inside its role
    const int rtid = rw * 64 + lane;                 // thread inside its role
    bool alive = true;
    FusedClock fclk;

    // Raw rows of (frame f, tile tl) -> LDS in 16-byte pieces, asynchronously (LDS-DMA) when DMA: two instructions
    // per wave and round.  (Dword pieces into padded rows -- conflict-free column reads -- took 72 LDS-DMA
    // instructions per workgroup and round, ~1.8 us of the CU's memory pipeline, and every poll of a team counter
    // queued behind them: the consumers saw `produced` 2.0 - 3.8 us late.  The 8-way bank conflicts of the unpadded
    // rows cost the sixteen ds_read_u16 of a round ~0.06 us.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 on the clobber list: meant
    auto stage_rows = [&](int f) {
        const uint8_t* const frame = stream + static_cast<size_t>(f) * (2 * N);
        constexpr int PPR = ROWB / 16;                      // pieces per row
        static_assert(N1 * PPR == 2 * kRoleThreads, "two pieces per producer thread");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * kRoleThreads + rtid;              // LDS piece q <- (row, piece of the row): fused_layout.h
            static_assert(Raw::PPR == PPR, "");
            const int row = Raw::row_of(q), piece = Raw::piece_of(q);
            const uint8_t* src = frame + 2 * (static_cast<size_t>(N2) * row + COLS * tl) + 16 * piece;
            if constexpr (DMA) {
                // In assembly: the compiler's wait-count bookkeeping then knows nothing of the LDS-DMA and puts no
                // s_waitcnt vmcnt(0) between here and the end of the round.  (Told of it -- the builtin -- it waits in
                // front of the first LDS access it cannot prove disjoint from the DMA's target: right behind the
                // issue, or in front of the next poll of an LDS flag, as its heuristics fall.)  The wait that counts is
                // drain_vector_memory() at the end of the round.  No other instruction of this kernel uses m0.
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    static_cast<unsigned>(reinterpret_cast<unsigned long long>(raw + 16 * (i * kRoleThreads + rw * 64))));
                if constexpr (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(dst) : "memory", "m0");
                else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
            } else {
                uint16_t h[8];                              // (any even stream address)
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = *reinterpret_cast<const uint16_t*>(src + 2 * k);
                uint4 v;
                v.x = h[0] | (static_cast<uint32_t>(h[1]) << 16);
                v.y = h[2] | (static_cast<uint32_t>(h[3]) << 16);
                v.z = h[4] | (static_cast<uint32_t>(h[5]) << 16);
                v.w = h[6] | (static_cast<uint32_t>(h[7]) << 16);
                *reinterpret_cast<uint4*>(raw + 16 * q) = v;
            }
        }
    };
#pragma clang diagnostic pop
    // (knob 1 = 7 / 8, measurement: the consumer / producer waves at raised issue priority)
    if (FKNOB(1) == 7 && !producer) __builtin_amdgcn_s_setprio(2);
    if (FKNOB(1) == 8 && producer) __builtin_amdgcn_s_setprio(2);
    if (producer) {
        // ================================ producers: columns ====================================
        const int sub0 = lane / TA, t0 = lane % TA;
        cf tw[GA::NPASS - 1][P - 1];                      // (pass 1 only: the later passes' come from twtabA)
        load_twiddles<GA, 1, true>(t0, tw_n1, tw);
        // This lane's columns never change.  The inter-step twiddle of (column c, register a) is W_N^{c bin_of(t, a)} and
        // bin_of(t, a) = bin_of(t, 0) + bin_of(0, a) for these geometries (the digits of 8 t + a do not carry): one
        // per-lane factor per column in a register, the eight per-register factors of the column in LDS -- 4 VGPRs
        // instead of 32 at the price of a second complex product per value.  (twN is in lane order: twN[c N1 + T a + t].)
        cf ustep[GROUPS];
        float wsgn[WINDOW ? GROUPS : 1][P];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int cl = (rw * S::SUBA + sub0) * GROUPS + g;     // a lane's two columns are neighbours: one dword of a raw row
            const int c = COLS * tl + cl;
            ustep[g] = twN[static_cast<size_t>(c) * N1 + t0];
            if (t0 < P) steptab[cl * P + t0] = twN[static_cast<size_t>(c) * N1 + TA * t0];
            static_assert(TA >= P, "one lane per register index");
#pragma unroll
            for (int a = 0; a < P; ++a)
                if constexpr (WINDOW) wsgn[g][a] = window[static_cast<size_t>(c) * N1 + t0 + TA * a] * ((c & 1) ? -1.0f : 1.0f);
        }
        exchange_sync<false>();              // (steptab rows are written and read by the same lane group)
        if (nj > 0 && xcd * FR + fsl < nframes) stage_rows(xcd * FR + fsl);
        unsigned nbar = 0;
        drain_vector_memory();
        alive = role_barrier(sy, 0, (nbar += kRoleWaves), lane);          // the first round's raw rows are in
        fclk.start();
#pragma unroll 1
        for (int j = 0; j < nj && alive; ++j) {
            const int f = (xcd + 8 * j) * FR + fsl;
            const bool valid = f < nframes;
            // Per-lane indices are re-derived every round from an opaque copy of the lane number: hoisted out of the loop
            // they -- and every LDS offset and store address built from them, all loop-invariant now that the team has
            // ONE buffer -- stay live across the round and spill at 128 VGPRs.
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const int sub = lane_ / TA, t = lane_ % TA;
            cf* const slab = slabsA + rw * S::SLAB_A + sub * GA::LDS_CPX;
            // Both column groups' samples go to registers first: the raw area is then free for the next round's rows,
            // whose LDS-DMA runs under this round's arithmetic.
            // One dword of a row holds the lane's sample of BOTH groups (columns 2 k, 2 k + 1: I0 Q0 I1 Q1): eight LDS reads
            // per thread and round instead of sixteen, and these are the kernel's worst -- consecutive lanes read
            // consecutive unpadded rows, 32 ... 128 bytes apart, 16 lanes to a bank (SQ_LDS_BANK_CONFLICT: a third of the
            // kernel's LDS cycles are conflicts, half of them here; profiles/r04_c4_fused.txt 6.).
            uint32_t iq[P];
            static_assert(GROUPS == 2, "a dword of raw bytes = two columns");
            if (valid) {
                const int cl0 = (rw * S::SUBA + sub) * GROUPS;
#pragma unroll
                for (int a = 0; a < P; ++a) {
                    const int row = t + TA * a, byte = 2 * cl0;                 // (stage_rows' layout)
                    iq[a] = *reinterpret_cast<const uint32_t*>(raw + Raw::offset(row, byte));
                }
            }
            if (!(alive = role_barrier(sy, 0, (nbar += kRoleWaves), lane))) break;
            if (j + 1 < nj && f + 8 * FR < nframes && FKNOB(1) < 3) stage_rows(f + 8 * FR);     // (knob 1 = 3: no raw rows -- garbage in, timing only)
            FSTAMP(0);                       // samples in registers, next rows on their way
            // Both column groups are transformed BEFORE the team's one buffer is free (the consumers are still loading
            // the previous round out of it): the second group waits in this wave's slab in natural k1 order, the
            // first in registers.
            cf y0[P];
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const int cl = (rw * S::SUBA + sub) * GROUPS + g;       // this lane group's column inside the tile
                const int c = COLS * tl + cl;                           // n2
                if (valid) {
                    const float sgn = (c & 1) ? -1.0f : 1.0f;           // (-1)^n, n = N2 n1 + n2
                    const float off = -(kTwo23 + 127.0f) * sgn;
                    cf x[P];
#pragma unroll
                    for (int a = 0; a < P; ++a) {
                        const cf v = g ? iq_pair_plus_2p23<1>(iq[a]) : iq_pair_plus_2p23<0>(iq[a]);
                        if constexpr (WINDOW) x[a] = (v - (kTwo23 + 127.0f)) * wsgn[g][a];
                        else x[a] = v * sgn + off;
                    }
                    if (!FKNOB(3)) group_fft_twlds<GA>(t, x, tw, slab, twtabA);
                    exchange_sync<false>();
                    if (g == 0) {
#pragma unroll
                        for (int a = 0; a < P; ++a) y0[a] = cmul(cmul(x[a], ustep[0]), steptab[cl * P + a]);
                    } else {
#pragma unroll
                        for (int a = 0; a < P; ++a)
                            slab[GA::slot(bin_of<GA>(t, a))] = cmul(cmul(x[a], ustep[g]), steptab[cl * P + a]);
                        exchange_sync<false>();
                    }
                }
            }
            FSTAMP(1);                       // both column groups transformed
            // (knob 1 = 5: the next rows' LDS-DMA is issued here instead, under the wait for the buffer; 6: and the wave keeps
            //  off the LDS until they have landed.  Both measured slower than the issue at the top of the round, 0.255 / 0.257
            //  against 0.285 Tsample/s: profiles/r04_c4_fused.txt.)
            if (FKNOB(1) >= 5 && j + 1 < nj && f + 8 * FR < nframes) stage_rows(f + 8 * FR);
            if (FKNOB(1) == 6 && j < NBUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (HALFF) {
                // ---- half-frame form: group 1 (in the slab) goes out first, half by half, as the two half buffers come free;
                // then group 0 -- and each half is signalled as soon as ITS stores have drained
                auto store_half = [&](int g, int h, int b) {
                    const int c = COLS * tl + (rw * S::SUBA + sub) * GROUPS + g;
                    cf* const ycol = Yteam + static_cast<size_t>(b) * HB + static_cast<size_t>(c) * RT;
#pragma unroll
                    for (int a = h * (P / 4); a < (h + 1) * (P / 4); ++a) {
                        const int e = 2 * t + 2 * TA * a;
                        cf4 v;
                        v.lo = slab[GA::slot(e)];
                        v.hi = slab[GA::slot(e + 1)];
                        *reinterpret_cast<cf4*>(ycol + static_cast<size_t>((e / RT) % HT) * (N2 * RT) + e % RT) = v;
                    }
                };
                static_assert(P % 4 == 0 && (2 * TA * (P / 4)) % RT == 0 && (2 * TA * (P / 4)) / RT == HT, "the store loop splits at k1 = N1 / 2");
                const int q0 = 2 * j, b0 = q0 % 3, b1 = (q0 + 1) % 3;
                const unsigned u0 = static_cast<unsigned>(q0 / 3), u1 = static_cast<unsigned>((q0 + 1) / 3);
                if (u0 > 0 && !(alive = team_wait(ctl, sy, &ctl->consumed[xcd][b0][0], &sy->seen[0][b0], 16u * u0, rw == 0, lane))) break;
                FTRACE(3);
                FSTAMP(2);                   // wait: first half buffer free
                if (valid) store_half(1, 0, b0);
                if (u1 > 0 && !(alive = team_wait(ctl, sy, &ctl->consumed[xcd][b1][0], &sy->seen[0][b1], 16u * u1, rw == 0, lane))) break;
                if (valid) {
                    store_half(1, 1, b1);
                    exchange_sync<false>();
#pragma unroll
                    for (int a = 0; a < P; ++a) slab[GA::slot(bin_of<GA>(t, a))] = y0[a];
                    exchange_sync<false>();
                    store_half(0, 0, b0);
                }
                drain_vector_memory();
                if (!(alive = role_barrier(sy, 0, (nbar += kRoleWaves), lane))) break;
                if (rw == 0 && lane == 0)
                    __hip_atomic_fetch_add(&ctl->produced[xcd][b0][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                FSTAMP(3);                   // first half stored, drained, signalled
                if (valid) {
                    store_half(0, 1, b1);
                    exchange_sync<false>();
                }
                drain_vector_memory();
                if (!(alive = role_barrier(sy, 0, (nbar += kRoleWaves), lane))) break;
                if (rw == 0 && lane == 0)
                    __hip_atomic_fetch_add(&ctl->produced[xcd][b1][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
            // the team has finished reading round j - NBUF out of the buffer
            if (j >= NBUF && !(alive = team_wait(ctl, sy, &ctl->consumed[xcd][j % NBUF][0], &sy->seen[0][j % NBUF], 32u * (j / NBUF),
                                                 rw == 0, lane, FKNOB(1) == 6))) break;
            FTRACE(3);
            FSTAMP(2);                       // wait: buffer free
            if (valid) {
#pragma unroll
                for (int g = GROUPS - 1; g >= 0; --g) {
                    const int c = COLS * tl + (rw * S::SUBA + sub) * GROUPS + g;
                    if (g == 0) {
#pragma unroll
                        for (int a = 0; a < P; ++a) slab[GA::slot(bin_of<GA>(t, a))] = y0[a];
                        exchange_sync<false>();
                    }
                    cf* const ycol = Yteam + static_cast<size_t>(j % NBUF) * FR * N + static_cast<size_t>(c) * RT;      // tile-major, like K2a
#pragma unroll
                    for (int a = 0; a < P / 2; ++a) {
                        const int e = 2 * t + 2 * TA * a;
                        cf4 v;
                        v.lo = slab[GA::slot(e)];
                        v.hi = slab[GA::slot(e + 1)];
                        *reinterpret_cast<cf4*>(ycol + static_cast<size_t>((e / RT + YROT) % TPF) * (N2 * RT) + e % RT) = v;
                    }
                    exchange_sync<false>();
                }
            }
            // The round's rows of Y sit in the team's L2 once every producer wave's stores have drained (plain stores:
            // the lines stay there, dirty -- profiles/r04_l2_residency.txt); the next round's raw rows landed long ago.
            drain_vector_memory();
            FSTAMP(3);                       // stores issued and drained
            if (!(alive = role_barrier(sy, 0, (nbar += kRoleWaves), lane))) break;
            if (rw == 0 && lane == 0)
                __hip_atomic_fetch_add(&ctl->produced[xcd][j % NBUF][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            FTRACE(0);
            FSTAMP(4);                       // arrived
        }
        fclk.publish(0, rw == 0 && lane == 0, 32 * xcd + rank);
    } else {
        // ================================ consumers: rows =======================================
        const int sub = lane / TB, t = lane % TB;
        cf* const slab = slabsB + rw * S::SLAB_B + sub * GB::LDS_CPX;
        cf tw[GB::NPASS - 1][P - 1];                      // (pass 1 only: the later passes' come from twtabB)
        load_twiddles<GB, 1, true>(t, tw_n2, tw);
        // (the accumulators belong to this branch alone: declared outside they would cost the producers 32 registers)
        double acc[GROUPS][P];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g)
#pragma unroll
            for (int a = 0; a < P; ++a) acc[g][a] = 0.0;
        constexpr int PERB = N2 * RT / kRoleThreads / 2;     // 16-byte loads per thread and tile
        typedef float f4 __attribute__((ext_vector_type(4)));
        unsigned nbar = 0;
        fclk.start();
#pragma unroll 1
        for (int j = 0; j < nj && alive; ++j) {
            const int f = (xcd + 8 * j) * FR + fsl;
            const bool valid = f < nframes;
            // (half-frame form: this workgroup's tile lies in half h of the round, half round q = 2 j + h, buffer q mod 3)
            const int hq = HALFF ? 2 * j + tl / (HT > 0 ? HT : 1) : j;
            const int cb = HALFF ? hq % 3 : j % NBUF;
            const unsigned cu = static_cast<unsigned>(HALFF ? hq / 3 : j / NBUF);
            if (!(alive = team_wait(ctl, sy, &ctl->produced[xcd][cb][0], &sy->seen[1][cb], 32u * (cu + 1), rw == 0, lane))) break;
            FTRACE(1);
            FSTAMP(0);                       // wait: round produced
            if (valid) {
                const cf* const yt = HALFF ? Yteam + static_cast<size_t>(cb) * HB + static_cast<size_t>(tl % (HT > 0 ? HT : 1)) * (N2 * RT)
                                           : Yteam + static_cast<size_t>(j % NBUF) * FR * N + static_cast<size_t>((tl + YROT) % TPF) * (N2 * RT);
                // sc1 loads: served by the L2, never by this CU's L1 (other CUs wrote these lines); 16 bytes per lane,
                // all eight of a thread's loads in flight -- and the wait for them in the SAME asm statement: the
                // compiler does not know that an asm load's destination is written when the data returns, and is free
                // to copy or reuse it before a separate s_waitcnt (round 4's L2 residency microbenchmark met exactly that: profiles/r04_l2_residency.txt)
                f4 v[PERB];
                static_assert(PERB == 8, "the tile is 8192 complex values");
                const cf* src[PERB];
#pragma unroll
                for (int i = 0; i < PERB; ++i) src[i] = yt + 2 * (i * kRoleThreads + rtid);
                if constexpr (NT == 1) {
                    asm volatile(
                        "global_load_dwordx4 %0, %8, off sc1 nt\n\t"
                        "global_load_dwordx4 %1, %9, off sc1 nt\n\t"
                        "global_load_dwordx4 %2, %10, off sc1 nt\n\t"
                        "global_load_dwordx4 %3, %11, off sc1 nt\n\t"
                        "global_load_dwordx4 %4, %12, off sc1 nt\n\t"
                        "global_load_dwordx4 %5, %13, off sc1 nt\n\t"
                        "global_load_dwordx4 %6, %14, off sc1 nt\n\t"
                        "global_load_dwordx4 %7, %15, off sc1 nt\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                        : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7])
                        : "memory");
                } else {
                    asm volatile(
                        "global_load_dwordx4 %0, %8, off sc1\n\t"
                        "global_load_dwordx4 %1, %9, off sc1\n\t"
                        "global_load_dwordx4 %2, %10, off sc1\n\t"
                        "global_load_dwordx4 %3, %11, off sc1\n\t"
                        "global_load_dwordx4 %4, %12, off sc1\n\t"
                        "global_load_dwordx4 %5, %13, off sc1\n\t"
                        "global_load_dwordx4 %6, %14, off sc1\n\t"
                        "global_load_dwordx4 %7, %15, off sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                        : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7])
                        : "memory");
                }
                // the previous round's columns have left the tile (arrivals posted long ago: see below)
                if (!(alive = role_wait(sy, 2, static_cast<unsigned>(kRoleWaves) * j))) break;
#pragma unroll
                for (int i = 0; i < PERB; ++i) {
                    const int idx = i * kRoleThreads + rtid;
                    cf* const dst = tile + (idx / HALF) * S::ROW_PITCH + 2 * (idx % HALF);
                    dst[0] = cf{v[i].x, v[i].y};
                    dst[1] = cf{v[i].z, v[i].w};
                }
            }
            FSTAMP(1);                       // tile loaded
            if (!(alive = role_barrier(sy, 1, (nbar += kRoleWaves), lane))) break;
            // every consumer wave's loads have returned: the team may overwrite the buffer
            if (rw == 0 && lane == 0)
                __hip_atomic_fetch_add(&ctl->consumed[xcd][cb][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            FTRACE(2);
            FSTAMP(2);
            if (valid) {
#pragma unroll
                for (int g = 0; g < GROUPS; ++g) {
                    const int jrow = (rw * GROUPS + g) * S::SUBB + sub;
                    cf x[P];
#pragma unroll
                    for (int a = 0; a < P; ++a) x[a] = tile[(t + TB * a) * S::ROW_PITCH + jrow];
                    // second group's column read: this wave is done with the tile.  Arrive now, nobody waits here; the
                    // next round's tile write waits for all eight arrivals, by then long posted -- a barrier behind the
                    // transforms, where the waves are skewed, cost 1.5 us per round
                    if (g == GROUPS - 1) role_arrive(sy, 2, lane);
                    if (!FKNOB(2)) group_fft_but_last<GB, true, fourstep_wide2<GB>(S::N)>(t, x, tw, slab, twtabB, widetab);
                    last_pass_accumulate<GB, fourstep_is_wide(S::N)>(x, acc[g]);
                    exchange_sync<false>();
                }
            } else {
                role_arrive(sy, 2, lane);
            }
            FSTAMP(3);                       // rows transformed
        }
        fclk.publish(8, rw == 0 && lane == 0, 32 * xcd + rank);
        // partial spectrum of (team, frame slot): the accumulators go through the tile area (the consumers' own: every
        // consumer wave has left its last round's column reads behind at that round's barrier) as [N2 k2][ROW_PITCH] doubles
        if (alive) {
            double* const stage = reinterpret_cast<double*>(tile);
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const int jrow = (rw * GROUPS + g) * S::SUBB + sub;
#pragma unroll
                for (int a = 0; a < P; ++a) stage[bin_of<GB>(t, a) * S::ROW_PITCH + jrow] = acc[g][a];
            }
        }
    }

    // ---- partial spectrum of (team, frame slot): rows of tile tl, streamed out by all sixteen waves ----------------
    __syncthreads();                                 // (waves that gave up have left: the barrier does not wait for them)
    if (__hip_atomic_load(&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
    const double* const stage = reinterpret_cast<const double*>(tile);                // [N2 k2][ROW_PITCH]
    double* const out = partial + (static_cast<size_t>(xcd) * FR + fsl) * N + RT * tl;
#pragma unroll
    for (int i = 0; i < N2 * RT / kWG; ++i) {
        const int idx = i * kWG + tid;
        const int k2 = idx / RT, jr = idx % RT;
        out[static_cast<size_t>(k2) * N1 + jr] = stage[k2 * S::ROW_PITCH + jr];
    }
}

// K3 companion: what became of the fused launch (launch_fused_verdict).
__global__ void fused_verdict_kernel(const FusedCtl* __restrict__ ctl, double* __restrict__ out, int N,
                                     unsigned* __restrict__ verdict, unsigned* __restrict__ aborts)
{
    if (ctl->abort[0] == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (verdict) *verdict = 1u;
        if (aborts) *aborts += 1u;               // (launches of one engine are stream-ordered: one writer at a time)
    }
    if (out == nullptr) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        out[i] = __builtin_nan("");
}

// Test double of "a CU held by someone else's kernel" (launch_fourstep_fused, fault = 2): one wavefront that owns
// 96 KB of LDS -- no fused workgroup fits beside it -- until the fused launch has raised its abort flag (or ~20 s).
__global__ __launch_bounds__(64) void fused_squatter_kernel(const FusedCtl* __restrict__ ctl, unsigned* __restrict__ running)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char squat[];
    if (threadIdx.x == 0) {
        squat[0] = 1;
        __hip_atomic_store(running, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (l2_read_scalar(&ctl->abort[0]) == 0 && wall_clock64() - t0 < 2000000000ull)      // 100 MHz: 20 s
            __builtin_amdgcn_s_sleep(32);
    }
}

#ifdef RPF_FUSED_PROFILE
}  // namespace
}  // namespace rpf
extern "C" int rpf_debug_fused_trace(unsigned long long* out)      // 256 x 64 x 4 values
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rpf::g_fused_trace), sizeof(unsigned long long) * 256 * 64 * 4) == hipSuccess ? 0 : 1;
}
extern "C" int rpf_debug_fused_profile_wg(unsigned long long* out)      // 256 x 16 values
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rpf::g_fused_prof_wg), sizeof(unsigned long long) * 256 * 16) == hipSuccess ? 0 : 1;
}
extern "C" int rpf_debug_fused_knobs(const int* four)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(rpf::g_fused_knob), four, sizeof(int) * 4) == hipSuccess ? 0 : 1;
}
extern "C" int rpf_debug_fused_profile(unsigned long long* out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rpf::g_fused_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(rpf::g_fused_prof), z, sizeof z);
    }
    return 0;
}
namespace rpf {
namespace {
#endif


// Large Bluestein path: even N in (4096, 131072] that is not a power of two,
// M = M1 x M2 = 2^ceil(log2(2N-1)) (bluestein_tables.h has the identity):
//   K2a (BLU)  a = (v - 127) g zero-padded to M; columns of FFT_M #1 -> Y[f][n2][k1]
//   this kernel, per row k1: A[k1 + M1 k2] = row transform of Y (FFT_M #1 done);
//        z = conj(A * bhat); the second FFT_M reads z[M1 m1 + m2] with (m1, m2) =
//        (k2, k1), so its column transform (over m1 = k2, M2 points) is over the
//        very values this lane group holds: through the slab into natural order,
//        transform again, times W_M^{k1 q1}, coalesced store of Y2[f][k1][q1];
//   K2b on the transposed split <M2, M1>: rows of FFT_M #2 over k1, |c[q1 + M2 q2]|^2
//        accumulated; bins >= N of the M convolution outputs are ignored by K3.
template <class S>
__global__ __launch_bounds__(kWG, 4) void bluestein_mid_kernel(const cf* __restrict__ Y, int nframes,
                                                              const cf* __restrict__ tw_sub,
                                                              const cf* __restrict__ bhat,
                                                              const cf* __restrict__ twM,
                                                              cf* __restrict__ Y2)
{
    using G = typename S::GB;
    constexpr int N1 = S::N1, N2 = S::N2, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const tile = reinterpret_cast<cf*>(smem);                                   // [N2][ROW_PITCH]
    cf* const slabs = tile + N2 * S::ROW_PITCH;                                     // [16][SLAB_B]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int sub = lane / T, t = lane % T;
    const int jrow = wave * S::SUBB + sub;
    cf* const slab = slabs + wave * S::SLAB_B + sub * G::LDS_CPX;

    cf tw[G::NPASS - 1][G::P - 1];
    load_twiddles<G, 1>(t, tw_sub, tw);

    const int ntasks = nframes * S::ROW_TILES;
#pragma unroll 1
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int f = task / S::ROW_TILES, ktile = task % S::ROW_TILES;
        const cf* const yf = Y + static_cast<size_t>(f) * S::N + S::ROW_TILE * ktile;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N2 * S::ROW_TILE / kWG / 2; ++i) {          // 16-byte loads
            const int idx = i * kWG + tid;
            const int n2 = idx / (S::ROW_TILE / 2), j = 2 * (idx % (S::ROW_TILE / 2));
            const cf4 v = *reinterpret_cast<const cf4*>(yf + static_cast<size_t>(n2) * N1 + j);
            tile[n2 * S::ROW_PITCH + j] = v.lo;
            tile[n2 * S::ROW_PITCH + j + 1] = v.hi;
        }
        __syncthreads();
        const int k1 = S::ROW_TILE * ktile + jrow;
        cf x[G::P];
#pragma unroll
        for (int a = 0; a < G::P; ++a) x[a] = tile[(t + T * a) * S::ROW_PITCH + jrow];
        group_fft<G>(t, x, tw, slab);
        exchange_sync<false>();
#pragma unroll
        for (int a = 0; a < G::P; ++a) {
            const int k2 = bin_of<G>(t, a);
            cf z = cmul(x[a], bhat[static_cast<size_t>(k1) * N2 + T * a + t]);
            z.y = -z.y;
            slab[G::slot(k2)] = z;
        }
        exchange_sync<false>();
        phase_fetch<G, 1>(t, x, slab);       // natural order; pass 1 rewrites exactly these slots
        group_fft<G>(t, x, tw, slab);
        exchange_sync<false>();
#pragma unroll
        for (int a = 0; a < G::P; ++a) {
            const int q1 = bin_of<G>(t, a);
            slab[G::slot(q1)] = cmul(x[a], twM[static_cast<size_t>(k1) * N2 + T * a + t]);
        }
        exchange_sync<false>();
        cf* const row = Y2 + (static_cast<size_t>(f) * N1 + k1) * N2;
#pragma unroll
        for (int a = 0; a < G::P / 2; ++a) {                              // 16-byte stores
            const int e = 2 * t + 2 * T * a;
            cf4 v;
            v.lo = slab[G::slot(e)];
            v.hi = slab[G::slot(e + 1)];
            *reinterpret_cast<cf4*>(row + e) = v;
        }
        exchange_sync<false>();
    }
}


// ---- host: tables in lane order --------------------------------------------
// out[r * LEN + T a + t] = master[(r * bin_of<G>(t, a) * step) mod master_len]: row r of the
// inter-step twiddles W^{r k} in the order the lanes of a Geom<LEN, 8> transform hold bin k.
template <class G>
void lane_ordered_rows(const cf* master, size_t master_len, size_t step, int rows, std::vector<cf>& out)
{
    out.resize(static_cast<size_t>(rows) * G::N);
    for (int r = 0; r < rows; ++r)
        for (int a = 0; a < G::P; ++a)
            for (int t = 0; t < G::T; ++t)
                out[static_cast<size_t>(r) * G::N + G::T * a + t] =
                    master[(static_cast<size_t>(r) * bin_of<G>(t, a) * step) % master_len];
}
// out[r * LEN + T a + t] = table[r + rows * bin_of<G>(t, a)]
template <class G>
void lane_ordered_cols(const cf* table, int rows, std::vector<cf>& out)
{
    out.resize(static_cast<size_t>(rows) * G::N);
    for (int r = 0; r < rows; ++r)
        for (int a = 0; a < G::P; ++a)
            for (int t = 0; t < G::T; ++t)
                out[static_cast<size_t>(r) * G::N + G::T * a + t] = table[r + static_cast<size_t>(rows) * bin_of<G>(t, a)];
}

// ---------------------------------------------------------------- dispatch --
using ColsFn = void (*)(const uint8_t*, int, const cf*, const cf*, const float*, cf*, int, const cf*);
using MidFn = void (*)(const cf*, int, const cf*, const cf*, const cf*, cf*);
using RowsFn = void (*)(const cf*, int, const cf*, double*, int);

using FusedFn = void (*)(const uint8_t*, int, const cf*, const cf*, const cf*, const float*, cf*, double*, FusedCtl*);
using RowsTableFn = void (*)(const cf*, size_t, size_t, int, std::vector<cf>&);
using ColsTableFn = void (*)(const cf*, int, std::vector<cf>&);

struct SplitInfo {
    int N, N1, N2, cols_lds, rows_lds, batch, groups, row_tiles;
    ColsFn cols[2][2];   // [window][dma]
    RowsFn rows;
    RowsTableFn step_twiddles;   // W_N^{n2 k1} in K2a's lane order
    FusedFn fused[2][2];         // [window][dma]
    int fused_lds, fused_fr;     // LDS bytes; frames per team round
#ifdef RPF_FUSED_PROFILE
    FusedFn fused_ab[4];         // measurement only (RPF_FUSED_MODE=0..3, rectangular + LDS-DMA): NBUF 2/2/1/2, NT both/none/none/raw rows only
#endif
};

template <int N1, int N2>
SplitInfo make_split()
{
    using S = Split<N1, N2>;
    return SplitInfo{S::N, N1, N2, S::COLS_LDS, S::ROWS_LDS, S::BATCH, S::GROUPS, S::ROW_TILES,
                     {{fourstep_cols_kernel<S, false, false>, fourstep_cols_kernel<S, false, true>},
                      {fourstep_cols_kernel<S, true, false>, fourstep_cols_kernel<S, true, true>}},
                     fourstep_rows_kernel<S, true>, lane_ordered_rows<typename S::GA>,
                     {{fourstep_fused_kernel<S, false, false>, fourstep_fused_kernel<S, false, true>},
                      {fourstep_fused_kernel<S, true, false>, fourstep_fused_kernel<S, true, true>}},
                     fused_lds_bytes<S>(), 262144 / S::N,
#ifdef RPF_FUSED_PROFILE
                     {fourstep_fused_kernel<S, false, true, 2, 1>, fourstep_fused_kernel<S, false, true, 2, 0>,
                      fourstep_fused_kernel<S, false, true, 1, 0>, fourstep_fused_kernel<S, false, true, 2, 2>}
#endif
    };
}

const SplitInfo kSplits[] = {
    make_split<128, 128>(),   // 16384
    make_split<256, 128>(),   // 32768
    make_split<256, 256>(),   // 65536
    make_split<512, 256>(),   // 131072
    make_split<512, 512>(),   // 262144 (config C4)
};

const SplitInfo* find_split(int N)
{
    for (const SplitInfo& s : kSplits)
        if (s.N == N) return &s;
    return nullptr;
}

// Large Bluestein: the three kernels for M = M1 x M2.
struct BluSplitInfo {
    int M, M1, M2, cols_lds, mid_lds, rows_lds, batch, groups, row_tiles, mid_tiles;
    ColsFn cols[2];   // [dma]
    MidFn mid;
    RowsFn rows;      // on the transposed split
    RowsTableFn step_twiddles;    // W_M^{n2 k1} in K2a's lane order (rows n2 < M2)
    RowsTableFn step_twiddles2;   // W_M^{k1 q1} in the mid kernel's lane order (rows k1 < M1)
    ColsTableFn kernel_spectrum;  // bhat[k1 + M1 k2] in the mid kernel's lane order
};

template <int M1, int M2>
BluSplitInfo make_blu_split()
{
    using S = Split<M1, M2>;
    using SR = Split<M2, M1>;
    return BluSplitInfo{S::N, M1, M2, S::COLS_LDS, S::ROWS_LDS, SR::ROWS_LDS, S::BATCH, SR::GROUPS,
                        SR::ROW_TILES, S::ROW_TILES,
                        {fourstep_cols_kernel<S, false, false, true>, fourstep_cols_kernel<S, false, true, true>},
                        bluestein_mid_kernel<S>, fourstep_rows_kernel<SR, false>, lane_ordered_rows<typename S::GA>,
                        lane_ordered_rows<typename S::GB>, lane_ordered_cols<typename S::GB>};
}

const BluSplitInfo kBluSplits[] = {
    make_blu_split<128, 128>(), make_blu_split<256, 128>(), make_blu_split<256, 256>(),
    make_blu_split<512, 256>(), make_blu_split<512, 512>(),
};

const BluSplitInfo* find_blu_split(int N)
{
    if (N <= 4096 || (N & 1) || (N & (N - 1)) == 0) return nullptr;
    int M = 16384;
    while (M < 2 * N - 1) M *= 2;
    for (const BluSplitInfo& s : kBluSplits)
        if (s.M == M) return &s;
    return nullptr;
}

}  // namespace

bool fourstep_supported(int N) { return find_split(N) != nullptr; }

size_t fourstep_scratch_bytes(int N)
{
    const SplitInfo* s = find_split(N);
    return s ? sizeof(cf) * static_cast<size_t>(s->N) * s->batch : 0;
}
size_t fourstep_scratch_bytes_per_frame(int N)
{
    const SplitInfo* s = find_split(N);
    return s ? sizeof(cf) * static_cast<size_t>(s->N) : 0;
}

int fourstep_partial_slots(int N)
{
    const SplitInfo* s = find_split(N);
    return s ? s->groups : 0;
}

int fourstep_sub_lengths(int N, int* n1, int* n2)
{
    const SplitInfo* s = find_split(N);
    if (!s) return 0;
    *n1 = s->N1;
    *n2 = s->N2;
    return 1;
}

void fourstep_tables(int N, const float* window, std::vector<cf>& step_tw, std::vector<float>& window_t)
{
    const SplitInfo* s = find_split(N);
    step_tw.clear();
    window_t.clear();
    if (!s) return;
    std::vector<cf> master;
    make_twiddles(N, master);
    s->step_twiddles(master.data(), master.size(), 1, s->N2, step_tw);       // n2 k1 < N
    if (window) {
        window_t.resize(N);
        for (int c = 0; c < s->N2; ++c)
            for (int n1 = 0; n1 < s->N1; ++n1)
                window_t[static_cast<size_t>(c) * s->N1 + n1] = window[static_cast<size_t>(s->N2) * n1 + c];
    }
}

hipError_t fourstep_prepare(int N, int device, LaunchInfo* li)
{
    const SplitInfo* s = find_split(N);
    if (!s) return hipErrorInvalidValue;
    hipError_t err;
    for (int w = 0; w < 2; ++w)
        for (int d = 0; d < 2; ++d) {
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->cols[w][d]),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, s->cols_lds);
            if (err != hipSuccess) return err;
        }
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->rows),
                              hipFuncAttributeMaxDynamicSharedMemorySize, s->rows_lds);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    li->grid = prop.multiProcessorCount;      // one 1024-thread workgroup per CU
    li->block = kWG;
    li->fpw = 1;
    li->lds_bytes = s->rows_lds;
    return hipSuccess;
}

hipError_t launch_fourstep(int N, bool window, bool use_dma, const uint8_t* d_stream, long nframes,
                           const cf* d_tw_n1, const cf* d_tw_n2, const cf* d_twN, const float* d_window,
                           cf* d_scratch, size_t scratch_bytes, double* d_partial, int max_grid, hipStream_t stream)
{
    const SplitInfo* s = find_split(N);
    if (!s || nframes < 1) return hipErrorInvalidValue;
    // frames per launch pair: what the scratch the engine has grown so far holds (at most the compile-time batch)
    const long batch = std::min<long>(s->batch, static_cast<long>(scratch_bytes / (sizeof(cf) * static_cast<size_t>(s->N))));
    if (batch < 1) return hipErrorInvalidValue;
    const int rows_grid = s->row_tiles * s->groups;
    const ColsFn cols = s->cols[window ? 1 : 0][use_dma ? 1 : 0];
    bool first = true;
    for (long done = 0; done < nframes; done += batch) {
        const int nb = static_cast<int>(std::min<long>(batch, nframes - done));
        const uint8_t* src = d_stream + static_cast<size_t>(done) * 2 * s->N;
        const int cols_grid = std::min(max_grid, nb * (s->N2 / kColTile));
        hipLaunchKernelGGL(cols, dim3(cols_grid), dim3(kWG), s->cols_lds, stream, src, nb, d_tw_n1, d_twN,
                           d_window, d_scratch, 0, static_cast<const cf*>(nullptr));
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(s->rows, dim3(rows_grid), dim3(kWG), s->rows_lds, stream, d_scratch, nb, d_tw_n2,
                           d_partial, first ? 1 : 0);
        err = hipGetLastError();
        if (err != hipSuccess) return err;
        first = false;
    }
    return hipSuccess;
}

// ---- fused four-step -----------------------------------------------------------
size_t fourstep_fused_scratch_bytes(int N)       // Y of two rounds per XCD: 8 x 2 x 2 MB
{
    const SplitInfo* s = find_split(N);
    return s ? sizeof(cf) * static_cast<size_t>(s->N) * s->fused_fr * 8 * 2 : 0;
}
int fourstep_fused_slots(int N)
{
    const SplitInfo* s = find_split(N);
    return s ? 8 * s->fused_fr : 0;
}
size_t fourstep_fused_ctl_bytes() { return sizeof(FusedCtl); }

hipError_t fourstep_fused_prepare(int N, int device, int* grid)
{
    const SplitInfo* s = find_split(N);
    if (!s) return hipErrorInvalidValue;
    if (s->fused_lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t err;
    for (int w = 0; w < 2; ++w)
        for (int d = 0; d < 2; ++d) {
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->fused[w][d]),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, s->fused_lds);
            if (err != hipSuccess) return err;
        }
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    // eight teams of 32: one workgroup per CU on a 256-CU, 8-XCD part, all of them resident
    if (prop.multiProcessorCount != 256) return hipErrorInvalidValue;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(s->fused[0][1]), kWG,
                                                       s->fused_lds);
    if (err != hipSuccess) return err;
    if (per_cu < 1) return hipErrorInvalidValue;
    *grid = 256;
    return hipSuccess;
}

// Frames [0, nframes) -> d_partial[8 * FR][N] (overwritten).  d_ctl: fourstep_fused_ctl_bytes() of
// device memory; d_scratch: fourstep_fused_scratch_bytes(N).  Follow K3 (d_skip = fourstep_fused_abort_word) with launch_fused_verdict.
hipError_t launch_fourstep_fused(int N, bool window, bool use_dma, const uint8_t* d_stream, long nframes,
                                 const cf* d_tw_n1, const cf* d_tw_n2, const cf* d_twN, const float* d_window,
                                 cf* d_scratch, double* d_partial, void* d_ctl, hipStream_t stream, int fault)
{
    const SplitInfo* s = find_split(N);
    if (!s || nframes < 1 || nframes > 0x7fffffffL) return hipErrorInvalidValue;
    hipError_t err = hipMemsetAsync(d_ctl, 0, sizeof(FusedCtl), stream);
    if (err != hipSuccess) return err;
    if (fault == 1) {
        // (tests) a 33rd arrival on XCD 0: the teams are not eight times 32, every workgroup leaves at once
        err = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&static_cast<FusedCtl*>(d_ctl)->arrivals[0][0]), 1, 1, stream);
        if (err != hipSuccess) return err;
    } else if (fault == 2) {
        // (tests) one CU is taken before the launch: 255 workgroups register, the 256th cannot start, the spins run out.
        // The squatter leaves when it sees this launch's abort flag, so the flag must be clear before it starts.
        // Squatter state per DEVICE, under a lock (two engines on two devices, or two threads, may arm the hook)
        struct Squat { hipStream_t stream = nullptr; unsigned* running = nullptr; };
        static std::mutex squat_mutex;
        static Squat squats[64];
        int squat_device = 0;
        if ((err = hipGetDevice(&squat_device)) != hipSuccess) return err;
        std::lock_guard<std::mutex> squat_lock(squat_mutex);
        hipStream_t& squat_stream = squats[squat_device & 63].stream;
        unsigned*& running = squats[squat_device & 63].running;
        // (the whole device idle: a kernel still running elsewhere would let this launch's workgroups in first)
        if ((err = hipDeviceSynchronize()) != hipSuccess) return err;
        if (!squat_stream && (err = hipStreamCreateWithFlags(&squat_stream, hipStreamNonBlocking)) != hipSuccess) return err;
        if (!running && (err = hipHostMalloc(reinterpret_cast<void**>(&running), 64, hipHostMallocMapped)) != hipSuccess) return err;
        *running = 0;
        constexpr int kSquatLds = 96 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_squatter_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kSquatLds);
        hipLaunchKernelGGL(fused_squatter_kernel, dim3(1), dim3(64), kSquatLds, squat_stream,
                           static_cast<const FusedCtl*>(d_ctl), running);
        if ((err = hipGetLastError()) != hipSuccess) return err;
        for (int i = 0; i < 20000000 && __atomic_load_n(running, __ATOMIC_ACQUIRE) == 0; ++i) std::this_thread::yield();
        if (__atomic_load_n(running, __ATOMIC_ACQUIRE) == 0) return hipErrorNotReady;      // (the squatter never started)
    }
    FusedFn fn = s->fused[window ? 1 : 0][use_dma ? 1 : 0];
#ifdef RPF_FUSED_PROFILE
    if (const char* mode = getenv("RPF_FUSED_MODE"); mode && !window && use_dma) {
        fn = s->fused_ab[atoi(mode) & 3];
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, s->fused_lds);
    }
#endif
    // One fused launch at a time per device: the kernel needs every CU for itself (one 1024-thread workgroup with all of
    // the LDS per CU), and two of them dispatched side by side from two streams -- two engines on one device -- could each
    // hold CUs the other is waiting for until both give up.  Each launch waits for the previous one's event.
    {
        int device = 0;
        if ((err = hipGetDevice(&device)) != hipSuccess) return err;
        static std::mutex chain_mutex;
        static hipEvent_t last_launch[64] = {};
        std::lock_guard<std::mutex> lock(chain_mutex);
        hipEvent_t& ev = last_launch[device & 63];
        if (ev) {
            if ((err = hipStreamWaitEvent(stream, ev, 0)) != hipSuccess) return err;
        } else if ((err = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
            ev = nullptr;
            return err;
        }
        hipLaunchKernelGGL(fn, dim3(256), dim3(kWG), s->fused_lds, stream, d_stream,
                           static_cast<int>(nframes), d_tw_n1, d_tw_n2, d_twN, d_window, d_scratch, d_partial,
                           static_cast<FusedCtl*>(d_ctl));
        if ((err = hipGetLastError()) != hipSuccess) return err;
        return hipEventRecord(ev, stream);
    }
}

const unsigned* fourstep_fused_abort_word(const void* d_ctl) { return &static_cast<const FusedCtl*>(d_ctl)->abort[0]; }

hipError_t launch_fused_verdict(const void* d_ctl, double* d_out, int N, unsigned* verdict, unsigned* aborts, hipStream_t stream)
{
    hipLaunchKernelGGL(fused_verdict_kernel, dim3(d_out ? 64 : 1), dim3(256), 0, stream, static_cast<const FusedCtl*>(d_ctl),
                       d_out, N, verdict, aborts);
    return hipGetLastError();
}

// Did the last fused launch give up?  (Synchronises with the stream.)
hipError_t fourstep_fused_aborted(const void* d_ctl, hipStream_t stream, bool* aborted)
{
    FusedCtl h;
    hipError_t err = hipMemcpyAsync(&h, d_ctl, sizeof h, hipMemcpyDeviceToHost, stream);
    if (err != hipSuccess) return err;
    if ((err = hipStreamSynchronize(stream)) != hipSuccess) return err;
    *aborted = h.abort[0] != 0;
    return hipSuccess;
}


// ---- large Bluestein path --------------------------------------------------
bool bigblu_supported(int N) { return find_blu_split(N) != nullptr; }

int bigblu_lengths(int N, int* M, int* m1, int* m2)
{
    const BluSplitInfo* s = find_blu_split(N);
    if (!s) return 0;
    *M = s->M;
    *m1 = s->M1;
    *m2 = s->M2;
    return 1;
}

// g (N) and bhat (M) are bluestein_tables.h's; outputs are what launch_bigblu takes.
void bigblu_tables(int N, const cf* g, const cf* bhat, std::vector<cf>& g_t, std::vector<cf>& bhat_t,
                   std::vector<cf>& step_tw, std::vector<cf>& step_tw2)
{
    const BluSplitInfo* s = find_blu_split(N);
    if (!s) return;
    g_t.assign(static_cast<size_t>(s->M), cf{0.0f, 0.0f});                   // zero past the frame
    for (int n = 0; n < N; ++n) g_t[static_cast<size_t>(n % s->M2) * s->M1 + n / s->M2] = g[n];
    s->kernel_spectrum(bhat, s->M1, bhat_t);
    std::vector<cf> master;
    make_twiddles(s->M, master);
    s->step_twiddles(master.data(), master.size(), 1, s->M2, step_tw);
    s->step_twiddles2(master.data(), master.size(), 1, s->M1, step_tw2);
}

size_t bigblu_scratch_bytes(int N)      // Y and Y2, one after the other
{
    const BluSplitInfo* s = find_blu_split(N);
    return s ? 2 * sizeof(cf) * static_cast<size_t>(s->M) * s->batch : 0;
}
size_t bigblu_scratch_bytes_per_frame(int N)
{
    const BluSplitInfo* s = find_blu_split(N);
    return s ? 2 * sizeof(cf) * static_cast<size_t>(s->M) : 0;
}

int bigblu_partial_slots(int N)
{
    const BluSplitInfo* s = find_blu_split(N);
    return s ? s->groups : 0;
}

hipError_t bigblu_prepare(int N, int device, LaunchInfo* li)
{
    const BluSplitInfo* s = find_blu_split(N);
    if (!s) return hipErrorInvalidValue;
    hipError_t err;
    for (int d = 0; d < 2; ++d) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->cols[d]),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, s->cols_lds);
        if (err != hipSuccess) return err;
    }
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->mid), hipFuncAttributeMaxDynamicSharedMemorySize,
                              s->mid_lds);
    if (err != hipSuccess) return err;
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(s->rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                              s->rows_lds);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    li->grid = prop.multiProcessorCount;
    li->block = kWG;
    li->fpw = 1;
    li->lds_bytes = s->cols_lds;
    return hipSuccess;
}

hipError_t launch_bigblu(int N, bool use_dma, const uint8_t* d_stream, long nframes, const cf* d_tw_m1,
                         const cf* d_tw_m2, const cf* d_step_tw, const cf* d_step_tw2, const cf* d_g_t,
                         const cf* d_bhat_t, cf* d_scratch, size_t scratch_bytes, double* d_partial, int max_grid,
                         hipStream_t stream)
{
    const BluSplitInfo* s = find_blu_split(N);
    if (!s || nframes < 1) return hipErrorInvalidValue;
    const long batch = std::min<long>(s->batch, static_cast<long>(scratch_bytes / (2 * sizeof(cf) * static_cast<size_t>(s->M))));
    if (batch < 1) return hipErrorInvalidValue;
    cf* const Y = d_scratch;
    cf* const Y2 = d_scratch + static_cast<size_t>(s->M) * batch;
    const int rows_grid = s->row_tiles * s->groups;
    bool first = true;
    for (long done = 0; done < nframes; done += batch) {
        const int nb = static_cast<int>(std::min<long>(batch, nframes - done));
        const uint8_t* src = d_stream + static_cast<size_t>(done) * 2 * N;
        const int cols_grid = std::min(max_grid, nb * (s->M2 / kColTile));
        hipLaunchKernelGGL(s->cols[use_dma ? 1 : 0], dim3(cols_grid), dim3(kWG), s->cols_lds, stream, src, nb,
                           d_tw_m1, d_step_tw, static_cast<const float*>(nullptr), Y, N, d_g_t);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        const int mid_grid = std::min(max_grid, nb * s->mid_tiles);
        hipLaunchKernelGGL(s->mid, dim3(mid_grid), dim3(kWG), s->mid_lds, stream, Y, nb, d_tw_m2, d_bhat_t, d_step_tw2,
                           Y2);
        if ((err = hipGetLastError()) != hipSuccess) return err;
        // second transform's rows: over k1 (M1 points), tw table of length M1
        hipLaunchKernelGGL(s->rows, dim3(rows_grid), dim3(kWG), s->rows_lds, stream, Y2, nb, d_tw_m1, d_partial,
                           first ? 1 : 0);
        if ((err = hipGetLastError()) != hipSuccess) return err;
        first = false;
    }
    return hipSuccess;
}

}  // namespace rpf
