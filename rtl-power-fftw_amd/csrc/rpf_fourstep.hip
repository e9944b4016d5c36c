// rpf_fourstep.hip -- gfx950 kernels for transform lengths that do not fit one
// workgroup's LDS (config C4 of BASELINE.json: N = 262144 bins).
//
// Four-step decomposition, N = N1*N2 = 512*512, n = 512 n1 + n2, k = k1 + 512 k2:
//
//   X[k1 + 512 k2] = sum_{n2} W_512^{n2 k2} * ( W_N^{n2 k1} * sum_{n1} x[512 n1 + n2] W_512^{n1 k1} )
//
//   K2a fourstep_cols_kernel  for a tile of 64 columns n2: the raw u8 rows (128 B =
//        one cache line per n1) are staged in LDS by dword LDS-DMA into rows padded to
//        33 dwords (conflict-free column reads); each wavefront runs 512-point
//        column FFTs (8 points per lane, two wave-local LDS exchanges, no
//        s_barrier), multiplies by W_N^{n2 k1} and writes Y[frame][n2][k1] as
//        coalesced 512-byte rows.  (-1)^n = (-1)^n2 is a per-column constant.
//   K2b fourstep_rows_kernel  workgroup (k1 tile of 16 rows, frame group): loads the
//        [512 n2][16 k1] tile of Y (one full cache line per n2), each wavefront owns
//        one row k1 for the whole launch: 512-point FFT over n2, |X|^2 into 8
//        per-lane f64 register accumulators; at the end the 16 rows of a tile leave
//        as whole 128-byte lines of the per-frame-group partial spectrum.
//   K3  (rpf_kernels.hip) sums the frame-group partials into pwr.
//
// HBM/L2 traffic per frame: 0.5 MB raw (algorithmic) + 2 MB Y written + 2 MB Y read
// + 2 MB of W_N twiddles (L2-resident table); frames are processed in batches
// whose Y scratch (128 MB) stays inside the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

using G1 = Geom<512, 8>;              // the 512-point sub-transform: one wavefront, 8 points per lane
constexpr int kN1 = 512, kN2 = 512, kNBig = kN1 * kN2;
constexpr int kWG = 1024, kWaves = kWG / 64;
constexpr int kColTile = 64;          // columns per K2a tile (128 raw bytes per row)
constexpr int kRowDwords = 33;        // 32 data dwords + 1 pad per staged raw row
constexpr int kRowTile = 16;          // k1 rows per K2b tile (128 B of Y per n2)
constexpr int kRowPitch = kRowTile + 1;
constexpr int kFrameGroups = 8;       // K2b: frames f = fg mod 8 share a workgroup's accumulators
constexpr int kSlab = G1::LDS_CPX;    // 576 complex per wavefront

constexpr int kColsLds = kN1 * kRowDwords * 4 + kWaves * kSlab * (int)sizeof(cf);            // 141312
constexpr int kRowsLds = kN2 * kRowPitch * (int)sizeof(cf) + kWaves * kSlab * (int)sizeof(cf);  // 143360

// 512-point FFT of the 8 values per lane (pass-1 layout: lane t holds elements
// t + 64 a); leaves X[bin_of<G1>(t, a)] in register a.
__device__ __forceinline__ void wave_fft512(int t, cf* x, const cf (&tw)[G1::NPASS - 1][G1::P - 1],
                                            cf* slab)
{
    middle_passes<G1, 1>(t, x, tw, slab);
    phase_fetch<G1, G1::NPASS>(t, x, slab);
    phase_last<G1>(x);
}

template <bool WINDOW, bool DMA>
__global__ __launch_bounds__(kWG, 4) void fourstep_cols_kernel(const uint8_t* __restrict__ stream,
                                                              int nframes,
                                                              const cf* __restrict__ tw512,
                                                              const cf* __restrict__ twN,
                                                              const float* __restrict__ window,
                                                              cf* __restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* const raw = smem;                                                 // [512][33] dwords
    cf* const slabs = reinterpret_cast<cf*>(smem + kN1 * kRowDwords * 4);      // [16][576]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
    cf* const slab = slabs + wave * kSlab;

    cf tw[G1::NPASS - 1][G1::P - 1];
    load_twiddles<G1, 1>(t, tw512, tw);

    const int ntasks = nframes * (kN2 / kColTile);
#pragma unroll 1
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int f = task / (kN2 / kColTile), ct = task % (kN2 / kColTile);
        const uint8_t* const frame = stream + static_cast<size_t>(f) * (2 * kNBig);

        __syncthreads();   // the previous tile has been consumed by every wave
        // stage the [512 rows][128 B] raw tile: LDS dword L <- row L/33, dword L%33
#pragma unroll 1
        for (int i = 0; i < (kN1 * kRowDwords + kWG - 1) / kWG; ++i) {
            const int L = i * kWG + tid;
            const int r = L / kRowDwords, d = L % kRowDwords;
            if (L < kN1 * kRowDwords && d < 32) {
                const uint8_t* src = frame + 2 * (static_cast<size_t>(kN2) * r + kColTile * ct) + 4 * d;
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
        for (int j = 0; j < kColTile / kWaves; ++j) {
            const int cl = wave + kWaves * j;           // column inside the tile
            const int c = kColTile * ct + cl;           // n2
            const float sgn = (c & 1) ? -1.0f : 1.0f;   // (-1)^n, n = 512 n1 + n2
            const float off = -(kTwo23 + 127.0f) * sgn;
            cf x[G1::P];
#pragma unroll
            for (int a = 0; a < G1::P; ++a) {
                const int n1 = t + 64 * a;
                const uint32_t iq =
                    *reinterpret_cast<const uint16_t*>(raw + 4 * (n1 * kRowDwords + (cl >> 1)) + 2 * (cl & 1));
                const cf v = iq_plus_2p23(iq);
                if constexpr (WINDOW) {
                    const float w = window[static_cast<size_t>(kN2) * n1 + c] * sgn;
                    x[a] = (v - (kTwo23 + 127.0f)) * w;
                } else {
                    x[a] = v * sgn + off;
                }
            }
            wave_fft512(t, x, tw, slab);
            // inter-step twiddle W_N^{n2 k1} (n2 k1 < N: no reduction needed), then
            // through the wave's slab into natural k1 order for a coalesced row store
            exchange_sync<false>();
#pragma unroll
            for (int a = 0; a < G1::P; ++a) {
                const int k1 = bin_of<G1>(t, a);
                slab[G1::slot(k1)] = cmul(x[a], twN[c * k1]);
            }
            exchange_sync<false>();
            cf* const yrow = Y + (static_cast<size_t>(f) * kN2 + c) * kN1;
#pragma unroll
            for (int a = 0; a < G1::P; ++a) yrow[t + 64 * a] = slab[G1::slot(t + 64 * a)];
            exchange_sync<false>();
        }
    }
}

__global__ __launch_bounds__(kWG, 4) void fourstep_rows_kernel(const cf* __restrict__ Y, int nframes,
                                                              const cf* __restrict__ tw512,
                                                              double* __restrict__ partial, int first)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const tile = reinterpret_cast<cf*>(smem);                                   // [512][17]
    cf* const slabs = tile + kN2 * kRowPitch;                                       // [16][576]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
    cf* const slab = slabs + wave * kSlab;
    const int ktile = blockIdx.x % (kN1 / kRowTile), fg = blockIdx.x / (kN1 / kRowTile);
    const int ngroups = gridDim.x / (kN1 / kRowTile);

    cf tw[G1::NPASS - 1][G1::P - 1];
    load_twiddles<G1, 1>(t, tw512, tw);
    double acc[G1::P];
#pragma unroll
    for (int a = 0; a < G1::P; ++a) acc[a] = 0.0;

#pragma unroll 1
    for (int f = fg; f < nframes; f += ngroups) {
        const cf* const yf = Y + static_cast<size_t>(f) * kNBig + kRowTile * ktile;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kN2 * kRowTile / kWG; ++i) {
            const int idx = i * kWG + tid;
            const int n2 = idx / kRowTile, j = idx % kRowTile;
            tile[n2 * kRowPitch + j] = yf[static_cast<size_t>(n2) * kN1 + j];
        }
        __syncthreads();
        cf x[G1::P];
#pragma unroll
        for (int a = 0; a < G1::P; ++a) x[a] = tile[(t + 64 * a) * kRowPitch + wave];
        wave_fft512(t, x, tw, slab);
        phase_accumulate(x, acc, G1::P);
        exchange_sync<false>();
    }

    // bins k = k1 + 512 k2: for a fixed k2 the tile's 16 rows are 16 consecutive
    // doubles = one 128-byte line of the partial spectrum
    __syncthreads();
    double* const stage = reinterpret_cast<double*>(smem);                          // [512 k2][17]
#pragma unroll
    for (int a = 0; a < G1::P; ++a) stage[bin_of<G1>(t, a) * kRowPitch + wave] = acc[a];
    __syncthreads();
    double* const out = partial + static_cast<size_t>(fg) * kNBig + kRowTile * ktile;
#pragma unroll
    for (int i = 0; i < kN2 * kRowTile / kWG; ++i) {
        const int idx = i * kWG + tid;
        const int k2 = idx / kRowTile, j = idx % kRowTile;
        double* p = out + static_cast<size_t>(k2) * kN1 + j;
        const double v = stage[k2 * kRowPitch + j];
        *p = first ? v : (*p + v);
    }
}

template <class K>
hipError_t set_lds(K kernel, int bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

bool fourstep_supported(int N) { return N == kNBig; }

size_t fourstep_scratch_bytes(int N) { return fourstep_supported(N) ? sizeof(cf) * kNBig * kFourStepBatch : 0; }

int fourstep_partial_slots(int N) { return fourstep_supported(N) ? kFrameGroups : 0; }

hipError_t fourstep_prepare(int N, int device, LaunchInfo* li)
{
    if (!fourstep_supported(N)) return hipErrorInvalidValue;
    hipError_t err;
    if ((err = set_lds(fourstep_cols_kernel<false, false>, kColsLds)) != hipSuccess) return err;
    if ((err = set_lds(fourstep_cols_kernel<false, true>, kColsLds)) != hipSuccess) return err;
    if ((err = set_lds(fourstep_cols_kernel<true, false>, kColsLds)) != hipSuccess) return err;
    if ((err = set_lds(fourstep_cols_kernel<true, true>, kColsLds)) != hipSuccess) return err;
    if ((err = set_lds(fourstep_rows_kernel, kRowsLds)) != hipSuccess) return err;
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, device)) != hipSuccess) return err;
    li->grid = prop.multiProcessorCount;      // one 1024-thread workgroup per CU
    li->block = kWG;
    li->fpw = 1;
    li->lds_bytes = kRowsLds;
    return hipSuccess;
}

hipError_t launch_fourstep(int N, bool window, bool use_dma, const uint8_t* d_stream, long nframes,
                           const cf* d_tw512, const cf* d_twN, const float* d_window, cf* d_scratch,
                           double* d_partial, int max_grid, hipStream_t stream)
{
    if (!fourstep_supported(N) || nframes < 1) return hipErrorInvalidValue;
    const int rows_grid = (kN1 / kRowTile) * kFrameGroups;
    bool first = true;
    for (long done = 0; done < nframes; done += kFourStepBatch) {
        const int nb = static_cast<int>(std::min<long>(kFourStepBatch, nframes - done));
        const uint8_t* src = d_stream + static_cast<size_t>(done) * 2 * kNBig;
        const int cols_grid = std::min(max_grid, nb * (kN2 / kColTile));
        if (window) {
            if (use_dma)
                hipLaunchKernelGGL((fourstep_cols_kernel<true, true>), dim3(cols_grid), dim3(kWG), kColsLds,
                                   stream, src, nb, d_tw512, d_twN, d_window, d_scratch);
            else
                hipLaunchKernelGGL((fourstep_cols_kernel<true, false>), dim3(cols_grid), dim3(kWG), kColsLds,
                                   stream, src, nb, d_tw512, d_twN, d_window, d_scratch);
        } else {
            if (use_dma)
                hipLaunchKernelGGL((fourstep_cols_kernel<false, true>), dim3(cols_grid), dim3(kWG), kColsLds,
                                   stream, src, nb, d_tw512, d_twN, d_window, d_scratch);
            else
                hipLaunchKernelGGL((fourstep_cols_kernel<false, false>), dim3(cols_grid), dim3(kWG), kColsLds,
                                   stream, src, nb, d_tw512, d_twN, d_window, d_scratch);
        }
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(fourstep_rows_kernel, dim3(rows_grid), dim3(kWG), kRowsLds, stream, d_scratch,
                           nb, d_tw512, d_partial, first ? 1 : 0);
        err = hipGetLastError();
        if (err != hipSuccess) return err;
        first = false;
    }
    return hipSuccess;
}

}  // namespace rpf
