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
// 256 MB of Y (measured: 64 MB batches 13 % slower, 128 MB 2 % slower -- launch
// tails, not Infinity-Cache residency, decide).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

constexpr int kWG = 1024, kWaves = kWG / 64;
constexpr int kColTile = 64;          // columns per K2a tile (128 raw bytes per row)
constexpr int kRowDwords = 33;        // 32 data dwords + 1 pad per staged raw row
constexpr size_t kScratchBytes = 256u << 20;

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
    static constexpr int ROWS_LDS = N2 * ROW_PITCH * (int)sizeof(cf) + kWaves * SLAB_B * (int)sizeof(cf);
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
            cf* const yrow = Y + (static_cast<size_t>(f) * N2 + c) * N1;
#pragma unroll
            for (int a = 0; a < G::P; ++a) yrow[t + T * a] = slab[G::slot(t + T * a)];
            exchange_sync<false>();
        }
    }
}

template <class S>
__global__ __launch_bounds__(kWG, 4) void fourstep_rows_kernel(const cf* __restrict__ Y, int nframes,
                                                              const cf* __restrict__ tw_sub,
                                                              double* __restrict__ partial, int first)
{
    using G = typename S::GB;
    constexpr int N1 = S::N1, N2 = S::N2, T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const tile = reinterpret_cast<cf*>(smem);                                   // [N2][ROW_PITCH]
    cf* const slabs = tile + N2 * S::ROW_PITCH;                                     // [16][SLAB_B]

    const int tid = threadIdx.x;
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
    constexpr int PER = N2 * S::ROW_TILE / kWG;
    cf nxt[PER];
    auto fetch = [&](int f) {
        const cf* const yf = Y + static_cast<size_t>(f) * S::N + S::ROW_TILE * ktile;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * kWG + tid;
            nxt[i] = yf[static_cast<size_t>(idx / S::ROW_TILE) * N1 + idx % S::ROW_TILE];
        }
    };
    if (fg < nframes) fetch(fg);
#pragma unroll 1
    for (int f = fg; f < nframes; f += ngroups) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * kWG + tid;
            tile[(idx / S::ROW_TILE) * S::ROW_PITCH + idx % S::ROW_TILE] = nxt[i];
        }
        __syncthreads();
        if (f + ngroups < nframes) fetch(f + ngroups);
        cf x[G::P];
#pragma unroll
        for (int a = 0; a < G::P; ++a) x[a] = tile[(t + T * a) * S::ROW_PITCH + jrow];
        group_fft<G>(t, x, tw, slab);
        phase_accumulate(x, acc, G::P);
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
        for (int i = 0; i < N2 * S::ROW_TILE / kWG; ++i) {
            const int idx = i * kWG + tid;
            const int n2 = idx / S::ROW_TILE, j = idx % S::ROW_TILE;
            tile[n2 * S::ROW_PITCH + j] = yf[static_cast<size_t>(n2) * N1 + j];
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
        for (int a = 0; a < G::P; ++a) row[t + T * a] = slab[G::slot(t + T * a)];
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

using RowsTableFn = void (*)(const cf*, size_t, size_t, int, std::vector<cf>&);
using ColsTableFn = void (*)(const cf*, int, std::vector<cf>&);

struct SplitInfo {
    int N, N1, N2, cols_lds, rows_lds, batch, groups, row_tiles;
    ColsFn cols[2][2];   // [window][dma]
    RowsFn rows;
    RowsTableFn step_twiddles;   // W_N^{n2 k1} in K2a's lane order
};

template <int N1, int N2>
SplitInfo make_split()
{
    using S = Split<N1, N2>;
    return SplitInfo{S::N, N1, N2, S::COLS_LDS, S::ROWS_LDS, S::BATCH, S::GROUPS, S::ROW_TILES,
                     {{fourstep_cols_kernel<S, false, false>, fourstep_cols_kernel<S, false, true>},
                      {fourstep_cols_kernel<S, true, false>, fourstep_cols_kernel<S, true, true>}},
                     fourstep_rows_kernel<S>, lane_ordered_rows<typename S::GA>};
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
                        bluestein_mid_kernel<S>, fourstep_rows_kernel<SR>, lane_ordered_rows<typename S::GA>,
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
                           cf* d_scratch, double* d_partial, int max_grid, hipStream_t stream)
{
    const SplitInfo* s = find_split(N);
    if (!s || nframes < 1) return hipErrorInvalidValue;
    const int rows_grid = s->row_tiles * s->groups;
    const ColsFn cols = s->cols[window ? 1 : 0][use_dma ? 1 : 0];
    bool first = true;
    for (long done = 0; done < nframes; done += s->batch) {
        const int nb = static_cast<int>(std::min<long>(s->batch, nframes - done));
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
                         const cf* d_bhat_t, cf* d_scratch, double* d_partial, int max_grid, hipStream_t stream)
{
    const BluSplitInfo* s = find_blu_split(N);
    if (!s || nframes < 1) return hipErrorInvalidValue;
    cf* const Y = d_scratch;
    cf* const Y2 = d_scratch + static_cast<size_t>(s->M) * s->batch;
    const int rows_grid = s->row_tiles * s->groups;
    bool first = true;
    for (long done = 0; done < nframes; done += s->batch) {
        const int nb = static_cast<int>(std::min<long>(s->batch, nframes - done));
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
