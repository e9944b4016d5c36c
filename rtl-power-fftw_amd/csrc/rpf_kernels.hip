// rpf_kernels.hip -- gfx950 (CDNA4) kernels of the power-spectrum engine.
//
// K1  fft_accum_kernel   fused  u8-IQ unpack -> (-1)^n -> window -> FFT -> |X|^2 (f64)
//                        replaces the body of Datastore::fftThread
//                        (/root/reference/src/datastore.cxx:66-89).
// K3  reduce_kernel      deterministic sum of the per-workgroup partial spectra
//                        into pwr[N] (Datastore::pwr, datastore.h:53).
//
// K1 layout.  One frame (N complex samples = 2N bytes of HBM) is owned by
// T = N/P threads holding P points each; a 256-thread (or T-thread, if larger)
// workgroup runs WG/T frames side by side and walks the stream persistently
// (frame f -> workgroup (f / FPW) mod grid).  Per frame:
//   1. the 2N raw bytes arrive in LDS by LDS-DMA (global_load_lds_dwordx4,
//      16 B per lane, fully coalesced, no VGPR round trip), issued one frame
//      ahead so the HBM latency hides under the previous frame's butterflies;
//   2. each thread picks its P samples (stride T) out of LDS with ds_read_u16,
//      converts (v_cvt_f32_ubyteN), removes the 127 offset, applies (-1)^n and
//      the window -- all exact except the single window rounding;
//   3. radix-P butterflies in registers, twiddles held in registers for the
//      whole kernel, one padded LDS exchange between passes (bank-conflict
//      free, fft_core.h); exchanges that stay inside a wavefront need no
//      s_barrier;
//   4. |X|^2 is added in double into P per-thread register accumulators that
//      live for the whole kernel; they are written once, at the end, as one
//      partial spectrum per frame slot.
// HBM traffic per frame is therefore exactly the 2N input bytes; the kernel is
// bound by VALU + LDS, not by HBM (DESIGN.md has the numbers).  No MFMA.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>

#include <algorithm>

#include "bluestein_tables.h"
#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

// Stage the raw bytes this wavefront will unpack in the iteration whose slot-0
// frame is `fb` (wave-local, a-major layout: fft_core.h raw_source).  P/8
// instructions per wave, each moving 64 lanes x 16 B = eight 128-byte rows.
// Frames past the end are clamped to the last frame (never accumulated) so that
// every iteration issues the same number of DMA instructions and the counted
// s_waitcnt vmcnt(N) at the top of the frame loop stays exact.
// IDX: the frame index type -- long in the single-acquisition kernel, int in the
// scan kernel (a hop's frames; one 64-bit multiply less per DMA instruction).
template <class G, bool DMA, typename IDX>
__device__ __forceinline__ void stage_raw(const uint8_t* __restrict__ stream, IDX fb, IDX nframes,
                                          uint8_t* wave_raw, int wave, int lane)
{
    constexpr int PIECES = G::P / 8;
    constexpr int FRAME_BYTES = 2 * G::N;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int j = i * 1024 + lane * 16;
        int slot, off;
        raw_source<G>(wave, j, &slot, &off);
        IDX f = fb + slot;
        f = f < nframes ? f : nframes - 1;
        const uint8_t* src = stream + static_cast<long>(f) * FRAME_BYTES + off;
        if constexpr (DMA) {
            // LDS address = wave-uniform base + 16 * lane (added by the hardware)
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wave_raw + i * 1024), 16, 0, 0);
        } else {
            *reinterpret_cast<uint4*>(wave_raw + j) = *reinterpret_cast<const uint4*>(src);
        }
    }
}

// ---- K1, one acquisition per launch ------------------------------------------------------
// Frame f -> workgroup (f / FPW) mod grid: at any moment the grid reads one contiguous window of
// the stream.  (The scan kernel below walks several acquisitions per launch; for a single one
// this plain form measured 1.2 us per launch faster on the same box -- A/B in
// profiles/r03_k1_fixed_cost.txt -- so rpf_accumulate / rpf_accumulate_device keep it.)
template <class G, int WG, int OCC, bool WINDOW, bool DMA, bool DBUF, int ACCB = 0, bool PF32 = false,
          int RAWD = 2, int ABL = 0, bool TWLDS = false>
__global__ __launch_bounds__(WG, OCC) void fft_accum_kernel(const uint8_t* __restrict__ stream,
                                                            long nframes,
                                                            const cf* __restrict__ twN,
                                                            const float* __restrict__ window,
                                                            double* __restrict__ partial)
{
    constexpr int P = G::P, T = G::T, N = G::N, NPASS = G::NPASS;
    constexpr int FPW = WG / T;
    constexpr int NSLAB = DBUF ? 2 : 1;
    constexpr bool BLOCK_SYNC = (T > 64);
    static_assert(WG % T == 0 && WG % 64 == 0, "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const slab_base = reinterpret_cast<cf*>(smem);                        // [NSLAB][FPW][LDS_CPX]
    uint8_t* const raw_base = smem + NSLAB * FPW * G::LDS_CPX * sizeof(cf);  // [WG/64][RAWD][128 P]

    const int tid = threadIdx.x;
    const int fs = tid / T, t = tid % T;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int RAW_SLOT = kRawChunk * P;           // bytes one wave stages per frame
    constexpr int PIECES = P / 8;                     // DMA instructions per wave per frame
    uint8_t* const wave_raw = raw_base + wave * (RAWD * RAW_SLOT);

    // First thing: get the first frames' bytes moving (HBM latency overlaps the
    // constant loads below).  RAWREG keeps the next frame's 2P bytes per lane in
    // VGPRs instead of issuing LDS-DMA (two global_load_dwordx4 cost a few issue
    // cycles, an LDS-DMA instruction ~100-200).
    const long stride = static_cast<long>(gridDim.x) * FPW;
    long fb = static_cast<long>(blockIdx.x) * FPW;
    if (fb < nframes) {
#pragma unroll
        for (int d = 0; d < RAWD; ++d)
            stage_raw<G, DMA, long>(stream, fb + d * stride, nframes, wave_raw + d * RAW_SLOT, wave, lane);
    }

    // Loop-invariant per-thread constants: twiddles, sign, window.
    cf tw[NPASS - 1][P - 1];
    load_twiddles<G, 1, TWLDS>(t, twN, tw);
    cf* const twtable = reinterpret_cast<cf*>(raw_base + (WG / 64) * RAWD * (kRawChunk * P));
    if constexpr (TWLDS) {
        fill_twlds<G, 1>(tid, WG, twN, twtable);
        exchange_sync<true>();
    }
    const float sgn = (t & 1) ? -1.0f : 1.0f;
    float wsgn[P];
    if constexpr (WINDOW) {
#pragma unroll
        for (int a = 0; a < P; ++a) wsgn[a] = window[t + T * a] * sgn;
    }
    // ACCB (tuning variants only): |X|^2 is first summed over ACCB frames in float32, then added to the double
    // accumulators.  1 .. 99: one float per bin (two v_fma_f32; round 3's variant 22); 100 + n (round 6, VERDICT r05 item
    // 4 i): PACKED, (re^2, im^2) kept apart in one register pair per bin -- ONE v_pk_fma_f32 per bin and frame -- over n
    // frames.  Both depart from the reference's "square and sum in double" (datastore.cxx:83-85); neither is shipped.
    constexpr bool ACCP = ACCB >= 100;
    constexpr int ACCN = ACCP ? ACCB - 100 : ACCB;
    double acc[P];
    float acc32[ACCB > 0 && !ACCP ? P : 1];
    cf acc32p[ACCP ? P : 1];
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = 0.0;
    if constexpr (ACCP) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc32p[a] = cf{0.0f, 0.0f};
    } else if constexpr (ACCB > 0) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc32[a] = 0.0f;
    }

    PhaseClock clk;
    clk.start();
    for (int it = 0; fb < nframes; fb += stride, ++it) {
        const bool active = (fb + fs) < nframes;
        cf* const slab = slab_base + ((DBUF ? (it & 1) : 0) * FPW + fs) * G::LDS_CPX;
        uint8_t* const ring_slot = wave_raw + (it % RAWD) * RAW_SLOT;
        cf x[P];

        {
            // this frame's bytes have landed: every iteration issues exactly PIECES DMA
            // instructions per wave, so all but the newest (RAWD-1) frames' worth are done
            if constexpr (DMA && !(ABL & 8))
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RAWD - 1) * PIECES) : "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 0);                   // waiting for the staged bytes
            phase_unpack<G, WINDOW>(ring_slot + 2 * lane, sgn, wsgn, x);
            // The slot is refilled next: its ds_read_u16 must have RETURNED first (a DMA
            // that hits in L2/MALL can land before queued LDS reads execute -- seen as
            // sporadic 1e-3 errors), so wait for this wave's LDS reads, not just issue.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 1);                   // unpack
            // the slot has been consumed: refill it with the frame RAWD iterations ahead
            if constexpr (!(ABL & 8))
                stage_raw<G, DMA, long>(stream, fb + RAWD * stride, nframes, ring_slot, wave, lane);
            RPF_STAMP(clk, 3);                   // DMA issue
        }

        // single slab: every wave must be done with the previous frame's slab
        if constexpr (!DBUF) exchange_sync<BLOCK_SYNC>();
        RPF_STAMP(clk, 2);                       // top-of-frame barrier
        middle_passes<G, 1, ABL, TWLDS>(t, x, tw, slab, clk, twtable);   // stamps 4J..4J+3
        if constexpr (!(ABL & 4)) phase_fetch<G, NPASS>(t, x, slab);
        asm volatile("" : "+v"(x[0]));
        RPF_STAMP(clk, 12);                      // last fetch
        if constexpr (!(ABL & 2)) phase_last<G>(x);
        RPF_STAMP(clk, 13);                      // last butterfly
        if constexpr (ACCP) {
            if (active) {
#pragma unroll
                for (int a = 0; a < P; ++a)
                    asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc32p[a]) : "v"(x[a]));
            }
            if ((it % ACCN) == ACCN - 1) {
#pragma unroll
                for (int a = 0; a < P; ++a) {
                    acc[a] += static_cast<double>(acc32p[a].x) + static_cast<double>(acc32p[a].y);
                    acc32p[a] = cf{0.0f, 0.0f};
                }
            }
        } else if constexpr (ACCB > 0) {
            if (active) {
#pragma unroll
                for (int a = 0; a < P; ++a)
                    acc32[a] = __builtin_fmaf(x[a].x, x[a].x, __builtin_fmaf(x[a].y, x[a].y, acc32[a]));
            }
            if ((it % ACCB) == ACCB - 1) {
#pragma unroll
                for (int a = 0; a < P; ++a) {
                    acc[a] += static_cast<double>(acc32[a]);
                    acc32[a] = 0.0f;
                }
            }
        } else if constexpr (ABL & 1) {
#pragma unroll
            for (int a = 0; a < P; ++a) asm volatile("" ::"v"(x[a]));
        } else {
            if (active) phase_accumulate(x, acc, P);
        }
        RPF_STAMP(clk, 14);                      // accumulate
    }
    clk.publish(lane);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) prefetches
    if constexpr (ACCP) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc[a] += static_cast<double>(acc32p[a].x) + static_cast<double>(acc32p[a].y);
    } else if constexpr (ACCB > 0) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc[a] += static_cast<double>(acc32[a]);
    }

    // One partial spectrum per workgroup (the FPW frame slots are summed here);
    // every workgroup writes its partial, zeros included.  The accumulators go
    // through LDS (now free) so that the bin-scattered registers leave as fully
    // coalesced 512-byte rows: stage at a padded bin index (one spare double per
    // 16, conflict-free for the stride-16 bin pattern of bin_of), then stream out.
    exchange_sync<true>();
    double* const stage = reinterpret_cast<double*>(smem);          // [FPW][N + N/16]
    constexpr int SN = N + N / 16;
    static_assert(sizeof(double) * SN <= sizeof(cf) * G::LDS_CPX + 2 * N, "stage fits the LDS");
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const int bin = bin_of<G>(t, a);
        stage[fs * SN + bin + (bin >> 4)] = acc[a];
    }
    exchange_sync<true>();
    if constexpr (PF32) {
        for (int bin = tid; bin < N; bin += WG) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < FPW; ++k) v += stage[k * SN + bin + (bin >> 4)];
            reinterpret_cast<float*>(partial)[static_cast<size_t>(blockIdx.x) * N + bin] = static_cast<float>(v);
        }
    } else {
        // two neighbouring bins per lane = one 16-byte store: an 8-byte-per-lane store tail is
        // issue-bound at ~7 B/clk/CU (MI355X_MICROARCH.md), and every workgroup ends in one
        typedef double d2 __attribute__((ext_vector_type(2)));
        for (int bin = 2 * tid; bin < N; bin += 2 * WG) {
            d2 v = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < FPW; ++k) {
                v.x += stage[k * SN + bin + (bin >> 4)];
                v.y += stage[k * SN + bin + 1 + (bin >> 4)];
            }
            *reinterpret_cast<d2*>(partial + static_cast<size_t>(blockIdx.x) * N + bin) = v;
        }
    }
}


// ---- K1 with the first pass on the matrix pipe (N = 4096 = 16 x 256, rectangular window) -----
// Pass 1 of the decimation-in-frequency transform is Y[r][m] = sum_a x[m + 256 a] W_16^{a r}: a
// 16-point DFT over a for each of the 256 columns m, on INTEGER data ((v - 127) is exact in f16).
// In real form that is a 32 x 32 matrix (rows (r, re/im), columns (a, re/im)) times the 32 x 256
// matrix of samples -- one v_mfma_f32_32x32x16_f16 pair per 32 columns.  The matrix entries
// (0, +-1, +-sqrt(1/2), +-cos(pi/8), +-sin(pi/8)) are split into three f16 terms hi + mid + lo
// (33 bits; every f16 x f16 product is exact in the f32 accumulator) that accumulate into the
// SAME f32 accumulator, scaled by 2^13 so that the lo term stays a normal f16; 2^-13, the
// (-1)^n of datastore.cxx:73 (= (-1)^m: 256 a is even) and the pass-1 twiddle W_4096^{m r}
// are one per-lane complex constant per output.  The matrix pipe runs beside the VALU; what
// the VALU keeps of pass 1 is one v_perm + one v_pk_add_f16 per sample and the twiddles.
// Lane (l & 31, hh = l >> 5) of wave w (of the frame's four) ends with the outputs
// r = 4 q + 2 hh + u (q < 4, u < 2) of the columns m = 64 w + 32 cb + (l & 31), cb < 2 -- the
// MFMA C/D layout, row = (reg & 3) + 8 (reg >> 2) + 4 hh, with the rows ordered (r, re/im) --
// and stores them where the VALU pass 1 would have: element 256 r + m of the padded slab.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));

__device__ const double kCos16[16] = {
    1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508978178,
    0.0, -0.38268343236508978178, -0.70710678118654752440, -0.92387953251128673848,
    -1.0, -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508978178,
    0.0, 0.38268343236508978178, 0.70710678118654752440, 0.92387953251128673848};

constexpr int kMfmaScaleLog2 = 13;

// the lane's eight A-operand values of K-step ks, term d (0 hi, 1 mid, 2 lo)
__device__ __forceinline__ void mfma_a_fragment(int lane, int ks, h8 (&out)[3])
{
    const int i = lane & 31, kb = lane >> 5;
    const int r = 4 * (i >> 3) + 2 * ((i >> 2) & 1) + ((i >> 1) & 1), c_out = i & 1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int k = 16 * ks + 8 * kb + t, a = k >> 1, c_in = k & 1;
        const int idx = (a * r) & 15;
        const double co = kCos16[idx], si = kCos16[(idx + 12) & 15];       // sin(x) = cos(x - pi/2)
        // Y = (cos - i sin)(xr + i xi): re = cos xr + sin xi, im = -sin xr + cos xi
        const double m = (c_out == c_in) ? co : (c_out == 0 ? si : -si);
        double v = m * static_cast<double>(1 << kMfmaScaleLog2);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const _Float16 h = static_cast<_Float16>(static_cast<float>(v));
            out[d][t] = h;
            v -= static_cast<double>(static_cast<float>(h));
        }
    }
}

// Measured (profiles/r03_mfma_first_pass.txt): exact to the bar (1.7e-7 from float64 truth, 4.8e-8 from
// the VALU kernel on config C2) and SLOWER, 65.0 against 53.5 us per C2 launch -- the twelve MFMAs
// of a wave-frame cost their full 12 x 32 cycles on top of the remaining VALU work wherever they
// are issued (in order: 65.0; a frame ahead, pinned between the packed-f32 butterflies of passes 2
// and 3: 69.3; between the f64 accumulate instructions: 64.9): on this part the matrix pipe does not
// run beside packed-f32 or f64 vector arithmetic, it displaces it.  Kept in the tuning build only.
template <class G, int WG, int OCC, int RAWD>
__global__ __launch_bounds__(WG, OCC) void fft_accum_mfma_kernel(const uint8_t* __restrict__ stream,
                                                                 long nframes,
                                                                 const cf* __restrict__ twN,
                                                                 const float* __restrict__ /*window*/,
                                                                 double* __restrict__ partial)
{
    constexpr int P = G::P, T = G::T, N = G::N, NPASS = G::NPASS;
    constexpr int FPW = WG / T;
    static_assert(N == 4096 && P == 16 && T == 256 && WG % T == 0 && NPASS == 3, "the 16 x 16 x 16 split of N = 4096");
    constexpr bool DMA = true, TWLDS = true;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const slab_base = reinterpret_cast<cf*>(smem);                 // [FPW][LDS_CPX]
    uint8_t* const raw_base = smem + FPW * G::LDS_CPX * sizeof(cf);    // [WG/64][RAWD][128 P]

    const int tid = threadIdx.x;
    const int fs = tid / T, t = tid % T;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int RAW_SLOT = kRawChunk * P;
    constexpr int PIECES = P / 8;
    uint8_t* const wave_raw = raw_base + wave * (RAWD * RAW_SLOT);

    const long stride = static_cast<long>(gridDim.x) * FPW;
    long fb = static_cast<long>(blockIdx.x) * FPW;
    if (fb < nframes) {
#pragma unroll
        for (int d = 0; d < RAWD; ++d)
            stage_raw<G, DMA, long>(stream, fb + d * stride, nframes, wave_raw + d * RAW_SLOT, wave, lane);
    }

    // loop-invariant: the matrix (A) fragments, the per-output constants, the later passes' twiddle table
    h8 amat[2][3];
    mfma_a_fragment(lane, 0, amat[0]);
    mfma_a_fragment(lane, 1, amat[1]);
    const int w4 = wave & 3, ml = lane & 31, hh = lane >> 5;
    cf tw1[2][8];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int reg = 0; reg < 8; ++reg) {
            const int m = 64 * w4 + 32 * cb + ml, r = 4 * (reg >> 1) + 2 * hh + (reg & 1);
            const float sc = ((m & 1) ? -1.0f : 1.0f) / static_cast<float>(1 << kMfmaScaleLog2);
            tw1[cb][reg] = twN[m * r] * sc;
        }
    }
    cf tw[NPASS - 1][P - 1];                                           // (pass >= 2: in the LDS table)
    cf* const twtable = reinterpret_cast<cf*>(raw_base + (WG / 64) * RAWD * RAW_SLOT);
    fill_twlds<G, 1>(tid, WG, twN, twtable);
    exchange_sync<true>();
    double acc[P];
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = 0.0;
    // this lane's pass-1 outputs in the slab: element 256 r + m -> slot 272 r + m + m / 16
    const int store_slot = G::slot(256 * (2 * hh) + 64 * w4 + ml);
    cf* const slab = slab_base + fs * G::LDS_CPX;

    PhaseClock clk;
    clk.start();
    for (int it = 0; fb < nframes; fb += stride, ++it) {
        const bool active = (fb + fs) < nframes;
        uint8_t* const ring_slot = wave_raw + (it % RAWD) * RAW_SLOT;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RAWD - 1) * PIECES) : "memory");
        exchange_sync<false>();
        // B operand: sample a = 8 ks + 4 hh + j of column 32 cb + ml sits at byte 128 a + 2 (32 cb + ml) of the
        // wave's raw slot; (I, Q) -> (1024 + I, 1024 + Q) as two f16 by one v_perm, minus 1151 -> (v - 127)
        // exactly (datastore.cxx:75)
        const uint8_t* const lane_raw = ring_slot + 2 * ml + 512 * hh;
        h8 bmat[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t iq = *reinterpret_cast<const uint16_t*>(lane_raw + kRawChunk * (8 * ks + j) + 64 * cb);
                    const uint32_t bits = __builtin_amdgcn_perm(0x64646464u, iq, 0x04010400u);
                    const h2 v = __builtin_bit_cast(h2, bits) - h2{(_Float16)1151.0f, (_Float16)1151.0f};
                    bmat[cb][ks][2 * j] = v.x;
                    bmat[cb][ks][2 * j + 1] = v.y;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot's reads have returned
        exchange_sync<false>();
        stage_raw<G, DMA, long>(stream, fb + RAWD * stride, nframes, ring_slot, wave, lane);

        f16x y[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) y[cb][k] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(amat[ks][d], bmat[cb][ks], y[cb], 0, 0, 0);

        // y = 2^13 x the first pass: twiddle, sign, scale -> the slab
        cf x[P];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int reg = 0; reg < 8; ++reg) {
                const int v = 4 * (reg >> 1) + 2 * (reg & 1);
                x[8 * cb + reg] = cmul(cf{y[cb][v], y[cb][v + 1]}, tw1[cb][reg]);
            }
        exchange_sync<true>();                                   // every wave is done with the previous frame's slab
        {
            cf* const p = slab + store_slot;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int reg = 0; reg < 8; ++reg) p[34 * cb + 272 * (4 * (reg >> 1) + (reg & 1))] = x[8 * cb + reg];
        }
        exchange_sync<true>();
        middle_passes<G, 2, 0, TWLDS>(t, x, tw, slab, clk, twtable);
        phase_fetch<G, NPASS>(t, x, slab);
        phase_last<G>(x);
        if (active) phase_accumulate(x, acc, P);
    }
    clk.publish(lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    exchange_sync<true>();
    double* const stage = reinterpret_cast<double*>(smem);
    constexpr int SN = N + N / 16;
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const int bin = bin_of<G>(t, a);
        stage[fs * SN + bin + (bin >> 4)] = acc[a];
    }
    exchange_sync<true>();
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int bin = 2 * tid; bin < N; bin += 2 * WG) {
        d2 v = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < FPW; ++k) {
            v.x += stage[k * SN + bin + (bin >> 4)];
            v.y += stage[k * SN + bin + 1 + (bin >> 4)];
        }
        *reinterpret_cast<d2*>(partial + static_cast<size_t>(blockIdx.x) * N + bin) = v;
    }
}

// ---- K1, a scan of several acquisitions per launch -----------------------------------------
__device__ __forceinline__ void write_lane(int& v, int uniform_value, int lane_const)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(uniform_value), "n"(lane_const));
}

struct HopLanes {
    int v_nframes, v_begin, v_bias;
    unsigned v_stream_lo, v_stream_hi;
    // Scalar loads at constant offsets (a few s_load_dwordx16 through the scalar cache, the path
    // every kernel argument takes), then one v_writelane per entry.
    __device__ __forceinline__ void load(const HopArgs& a)
    {
        int nf = 0, bg = 0, bs = 0, lo = 0, hi = 0;
#pragma unroll
        for (int h = 0; h < kMaxHops; ++h) {
            const uintptr_t p = reinterpret_cast<uintptr_t>(a.stream[h]);
            write_lane(nf, a.nframes[h], h);
            write_lane(bg, a.it_begin[h], h);
            write_lane(bs, a.slot_bias[h], h);
            write_lane(lo, static_cast<int>(static_cast<unsigned>(p)), h);
            write_lane(hi, static_cast<int>(static_cast<unsigned>(p >> 32)), h);
        }
        write_lane(bg, a.it_begin[kMaxHops], kMaxHops);
        v_nframes = nf;
        v_begin = bg;
        v_bias = bs;
        v_stream_lo = static_cast<unsigned>(lo);
        v_stream_hi = static_cast<unsigned>(hi);
    }
    __device__ __forceinline__ int it_begin(int h) const { return __builtin_amdgcn_readlane(v_begin, h); }
    __device__ __forceinline__ int nframes(int h) const { return __builtin_amdgcn_readlane(v_nframes, h); }
    __device__ __forceinline__ int slot_bias(int h) const { return __builtin_amdgcn_readlane(v_bias, h); }
    __device__ __forceinline__ const uint8_t* stream(int h) const
    {
        const uintptr_t lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v_stream_lo), h));
        const uintptr_t hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v_stream_hi), h));
        return reinterpret_cast<const uint8_t*>(lo | (hi << 32));
    }
    // number of hop starts 1 .. kMaxHops at or before `it` (HopArgsView::hop_of)
    __device__ __forceinline__ int hop_of(int it) const
    {
        const int lane = static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(lane >= 1 && lane <= kMaxHops && v_begin <= it);
        return __builtin_popcountll(m);
    }
};

// DBUF: two slabs used alternately.  With one slab the pass-1 store of frame f+1
// must wait (workgroup barrier at the top of the loop) until every wave has
// finished reading frame f's slab; with two, the single barrier after the pass-1
// store orders both hazards and a frame costs one s_barrier instead of two, at
// the price of LDS (fewer resident workgroups).
//
// RAWD: depth of the raw-byte ring = iterations staged ahead by LDS-DMA.  The HBM
// latency seen by a DMA under load is several frame times (measured ~3 us vs
// ~1 us of butterflies per frame), so one frame ahead leaves the workgroup idle
// most of the time; RAWD iterations ahead keep RAWD x 2N bytes per frame slot in flight.
// ACCB > 0 (tuning variants): |X|^2 is first summed over ACCB frames in packed
// float32 (one v_pk_fma_f32 per bin instead of four half-rate f64 instructions)
// and only then folded into the f64 accumulators -- adds <= ~1e-7 relative error
// per batch, averaged down over the batches.  PF32: partial spectra leave as
// float32 (half the flush and K3 traffic; each partial is a sum over ~13 frames
// and there are hundreds of them, so the rounding averages out to ~1e-9).
//
// One launch walks the hops of `hops` (a single acquisition is H = 1): workgroup w owns the
// iterations [w I / G, (w + 1) I / G) of the launch's sequence and writes one partial spectrum
// per hop it touched (slot = slot_bias[h] + w), zeroing its register accumulators in between
// -- the reference's per-hop reset (acquisition.cxx:252-254) without a kernel boundary.
template <class G, int WG, int OCC, bool WINDOW, bool DMA, bool DBUF, int ACCB = 0, bool PF32 = false,
          int RAWD = 2, int ABL = 0, bool TWLDS = false>
__global__ __launch_bounds__(WG, OCC) void fft_accum_scan_kernel(const cf* __restrict__ twN,
                                                            const float* __restrict__ window,
                                                            double* __restrict__ partial,
                                                            const HopArgs hops)
{
    constexpr int P = G::P, T = G::T, N = G::N, NPASS = G::NPASS;
    constexpr int FPW = WG / T;
    constexpr int NSLAB = DBUF ? 2 : 1;
    constexpr bool BLOCK_SYNC = (T > 64);
    static_assert(WG % T == 0 && WG % 64 == 0, "");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const slab_base = reinterpret_cast<cf*>(smem);                        // [NSLAB][FPW][LDS_CPX]
    uint8_t* const raw_base = smem + NSLAB * FPW * G::LDS_CPX * sizeof(cf);  // [WG/64][RAWD][128 P]

    const int tid = threadIdx.x;
    const int fs = tid / T, t = tid % T;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int RAW_SLOT = kRawChunk * P;           // bytes one wave stages per frame
    constexpr int PIECES = P / 8;                     // DMA instructions per wave per frame
    uint8_t* const wave_raw = raw_base + wave * (RAWD * RAW_SLOT);

    // This workgroup's iterations: `count` of them, `step` apart from `first` on (hop_partition.h).
    // The host launches at most one workgroup per iteration (launch_fft_accum checks it), so
    // count >= 1 -- deliberately not tested here: a branch on q and r would put their scalar
    // load in front of the table loads instead of beside them.
    HopLanes tbl;
    tbl.load(hops);
    const int step = hops.step;
    int first, count;
    hop_share(static_cast<int>(blockIdx.x), hops.q, hops.r, step, &first, &count);

    // First thing: get the first iterations' bytes moving (HBM latency overlaps the constant
    // loads below).  The staging cursor `ahead` runs RAWD iterations in front of the compute
    // cursor, across hop boundaries; every iteration issues PIECES DMAs.  The frame loop sees
    // of it only a frame index that advances and a countdown: `ahead_run` stagings stay inside
    // the hop the cursor stands in (scans have step = 1, an interleaved single acquisition
    // never leaves its hop), then ahead_turn() moves the cursor on -- or, when nothing is left
    // to stage, parks it on the launch's iteration 0 with no advance: the same 2N FPW bytes
    // for every workgroup, an L2 hit, so the surplus (never read) stagings that keep the DMA
    // count per iteration constant cost no memory traffic.
    const int fstep = FPW * step;                  // frames between a workgroup's iterations
    HopCursor ahead;
    ahead.seek(tbl, first);
    int ahead_fb = (ahead.j - ahead.begin) * FPW, ahead_fstep = fstep;
    int ahead_left = count;                        // real iterations not staged yet
    auto run_length = [&](const HopCursor& c, int left) {
        const int in_hop = step == 1 ? c.end - c.j : left;
        return in_hop < left ? in_hop : left;
    };
    int ahead_run = run_length(ahead, ahead_left);
    ahead_left -= ahead_run;
    auto ahead_turn = [&]() {
        if (ahead_left > 0) {
            ahead.seek(tbl, ahead.end);            // (step == 1 here: the next hop starts where this one ended)
            ahead_fb = 0;
            ahead_run = run_length(ahead, ahead_left);
            ahead_left -= ahead_run;
        } else {
            ahead.seek(tbl, 0);
            ahead_fb = 0;
            ahead_fstep = 0;
            ahead_run = 0x7fffffff;
        }
    };
    auto stage_next = [&](uint8_t* dst) {
        stage_raw<G, DMA, int>(ahead.stream, ahead_fb, ahead.nframes, dst, wave, lane);
        ahead_fb += ahead_fstep;
        if (--ahead_run == 0) ahead_turn();
    };
    if constexpr (!(ABL & 8)) {
#pragma unroll
        for (int d = 0; d < RAWD; ++d) stage_next(wave_raw + d * RAW_SLOT);
    }

    // Loop-invariant per-thread constants: twiddles, sign, window.
    cf tw[NPASS - 1][P - 1];
    load_twiddles<G, 1, TWLDS>(t, twN, tw);
    cf* const twtable = reinterpret_cast<cf*>(raw_base + (WG / 64) * RAWD * (kRawChunk * P));
    if constexpr (TWLDS) {
        fill_twlds<G, 1>(tid, WG, twN, twtable);
        exchange_sync<true>();
    }
    const float sgn = (t & 1) ? -1.0f : 1.0f;
    float wsgn[P];
    if constexpr (WINDOW) {
#pragma unroll
        for (int a = 0; a < P; ++a) wsgn[a] = window[t + T * a] * sgn;
    }
    double acc[P];
    float acc32[ACCB > 0 ? P : 1];

    PhaseClock clk;
    clk.start();
    HopCursor cur;
    cur.seek(tbl, first);
    int it = 0;                                    // iterations done: ring slot and slab parity
    while (true) {
        // ---- one segment: this workgroup's iterations inside hop cur.h ------------------------
        const int seg = run_length(cur, count - it);
#pragma unroll
        for (int a = 0; a < P; ++a) acc[a] = 0.0;
        if constexpr (ACCB > 0) {
#pragma unroll
            for (int a = 0; a < P; ++a) acc32[a] = 0.0f;
        }
        int fb = (cur.j - cur.begin) * FPW;        // slot-0 frame of the iteration, within the hop
        for (int n = seg; n > 0; --n, ++it, fb += fstep) {
            const bool active = (fb + fs) < cur.nframes;
            cf* const slab = slab_base + ((DBUF ? (it & 1) : 0) * FPW + fs) * G::LDS_CPX;
            uint8_t* const ring_slot = wave_raw + (it % RAWD) * RAW_SLOT;
            cf x[P];

            // this iteration's bytes have landed: every iteration issues exactly PIECES DMA
            // instructions per wave, so all but the newest (RAWD-1) iterations' worth are done
            if constexpr (DMA && !(ABL & 8))
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RAWD - 1) * PIECES) : "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 0);                   // waiting for the staged bytes
            phase_unpack<G, WINDOW>(ring_slot + 2 * lane, sgn, wsgn, x);
            // The slot is refilled next: its ds_read_u16 must have RETURNED first (a DMA
            // that hits in L2/MALL can land before queued LDS reads execute -- seen as
            // sporadic 1e-3 errors), so wait for this wave's LDS reads, not just issue.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 1);                   // unpack
            // the slot has been consumed: refill it with the iteration RAWD ahead
            if constexpr (!(ABL & 8)) stage_next(ring_slot);
            RPF_STAMP(clk, 3);                   // DMA issue

            // single slab: every wave must be done with the previous frame's slab
            if constexpr (!DBUF) exchange_sync<BLOCK_SYNC>();
            RPF_STAMP(clk, 2);                       // top-of-frame barrier
            middle_passes<G, 1, ABL, TWLDS>(t, x, tw, slab, clk, twtable);   // stamps 4J..4J+3
            if constexpr (!(ABL & 4)) phase_fetch<G, NPASS>(t, x, slab);
            asm volatile("" : "+v"(x[0]));
            RPF_STAMP(clk, 12);                      // last fetch
            if constexpr (!(ABL & 2)) phase_last<G>(x);
            RPF_STAMP(clk, 13);                      // last butterfly
            if constexpr (ACCB > 0) {
                if (active) {
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        acc32[a] = __builtin_fmaf(x[a].x, x[a].x, __builtin_fmaf(x[a].y, x[a].y, acc32[a]));
                }
                if ((it % ACCB) == ACCB - 1) {
#pragma unroll
                    for (int a = 0; a < P; ++a) {
                        acc[a] += static_cast<double>(acc32[a]);
                        acc32[a] = 0.0f;
                    }
                }
            } else if constexpr (ABL & 1) {
#pragma unroll
                for (int a = 0; a < P; ++a) asm volatile("" ::"v"(x[a]));
            } else {
                if (active) phase_accumulate(x, acc, P);
            }
            RPF_STAMP(clk, 14);                      // accumulate
        }
        if constexpr (ACCB > 0) {
#pragma unroll
            for (int a = 0; a < P; ++a) acc[a] += static_cast<double>(acc32[a]);
        }

        // ---- hand the segment over: one partial spectrum (the FPW frame slots summed) ----------
        // The accumulators go through the slab (free between frames; the raw ring with its
        // in-flight prefetches is not touched) so that the bin-scattered registers leave as
        // fully coalesced 512-byte rows: stage at a padded bin index (one spare double per
        // 16, conflict-free for the stride-16 bin pattern of bin_of), then stream out.
        exchange_sync<true>();
        double* const stage = reinterpret_cast<double*>(smem);          // [FPW][N + N/16]
        constexpr int SN = N + N / 16;
        static_assert(sizeof(double) * SN <= sizeof(cf) * G::LDS_CPX, "the stage stays inside the slab");
        // (opaque copies of the thread indices: the hand-over runs once per hop, its sixteen
        // stage addresses must not be hoisted into registers that live across the frame loop)
        int ft = t, ftid = tid, ffs = fs;
        asm volatile("" : "+v"(ft), "+v"(ftid), "+v"(ffs));
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int bin = bin_of<G>(ft, a);
            stage[ffs * SN + bin + (bin >> 4)] = acc[a];
        }
        exchange_sync<true>();
        const size_t slot = static_cast<size_t>(tbl.slot_bias(cur.h) + static_cast<int>(blockIdx.x));
        if constexpr (PF32) {
            for (int bin = ftid; bin < N; bin += WG) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < FPW; ++k) v += stage[k * SN + bin + (bin >> 4)];
                reinterpret_cast<float*>(partial)[slot * N + bin] = static_cast<float>(v);
            }
        } else {
            // two neighbouring bins per lane = one 16-byte store: an 8-byte-per-lane store tail is
            // issue-bound at ~7 B/clk/CU (MI355X_MICROARCH.md), and every workgroup ends in one
            typedef double d2 __attribute__((ext_vector_type(2)));
            for (int bin = 2 * ftid; bin < N; bin += 2 * WG) {
                d2 v = {0.0, 0.0};
#pragma unroll
                for (int k = 0; k < FPW; ++k) {
                    v.x += stage[k * SN + bin + (bin >> 4)];
                    v.y += stage[k * SN + bin + 1 + (bin >> 4)];
                }
                *reinterpret_cast<d2*>(partial + slot * N + bin) = v;
            }
        }
        if (it >= count) break;
        // next hop: the slab is reused by its first frame once every wave has read the stage
        exchange_sync<true>();
        cur.seek(tbl, cur.end);                    // (a segment that is not the last ends with its hop)
    }
    clk.publish(lane);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (repeated) prefetches
}

// KB  bluestein_kernel -- any even N <= 4096 that is not one of K1's powers of two
// (bluestein_tables.h).  G = Geom<M, P> with M = 2^ceil(log2(2N-1)); per frame:
//   a[n] = (v[n] - 127) * g[n] for n < N, zero-padded to M      (g carries (-1)^n, window, chirp)
//   A = FFT_M(a);  z = conj(A * bhat);  c = FFT_M(z)            (= conj of the circular convolution)
//   pwr[k] += |c[k]|^2 for k < N                                (|X[k]| = |c[k]|)
// The two transforms reuse K1's passes; between them the spectrum goes through
// the slab once more (digit-reversed -> natural order).  Samples are read
// straight from HBM as coalesced u16 loads (frames are only 4-byte aligned).
template <class G, int WG, int OCC, bool TWLDS>
__global__ __launch_bounds__(WG, OCC) void bluestein_kernel(const uint8_t* __restrict__ stream,
                                                            long nframes, int N,
                                                            const cf* __restrict__ twM,
                                                            const cf* __restrict__ g,
                                                            const cf* __restrict__ bhat,
                                                            double* __restrict__ partial)
{
    constexpr int P = G::P, T = G::T, M = G::N, NPASS = G::NPASS;
    constexpr int FPW = WG / T;
    constexpr bool BLOCK_SYNC = (T > 64);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int fs = tid / T, t = tid % T;
    cf* const slab = reinterpret_cast<cf*>(smem) + fs * G::LDS_CPX;

    cf tw[NPASS - 1][P - 1];
    load_twiddles<G, 1, TWLDS>(t, twM, tw);
    cf* const twtable = reinterpret_cast<cf*>(smem) + FPW * G::LDS_CPX;
    if constexpr (TWLDS) {
        fill_twlds<G, 1>(tid, WG, twM, twtable);
        exchange_sync<true>();
    }
    PhaseClock noclk;
    double acc[P];
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = 0.0;

    const long stride = static_cast<long>(gridDim.x) * FPW;
    for (long fb = static_cast<long>(blockIdx.x) * FPW; fb < nframes; fb += stride) {
        const bool active = (fb + fs) < nframes;
        const uint8_t* const frame = stream + (fb + fs) * 2 * static_cast<long>(N);
        cf x[P];
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int n = t + T * a;
            x[a] = cf{0.0f, 0.0f};
            if (active && n < N) {
                const uint32_t iq = *reinterpret_cast<const uint16_t*>(frame + 2 * n);
                const cf v = iq_plus_2p23(iq) - (kTwo23 + 127.0f);
                x[a] = cmul(v, g[n]);
            }
        }
        exchange_sync<BLOCK_SYNC>();             // previous frame's slab reads are done
        middle_passes<G, 1, 0, TWLDS>(t, x, tw, slab, noclk, twtable);
        phase_fetch<G, NPASS>(t, x, slab);
        phase_last<G>(x);
        exchange_sync<BLOCK_SYNC>();             // ... before the slab is rewritten in another order
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int j = bin_of<G>(t, a);
            cf z = cmul(x[a], bhat[j]);
            z.y = -z.y;
            slab[G::slot(j)] = z;
        }
        exchange_sync<BLOCK_SYNC>();
        phase_fetch<G, 1>(t, x, slab);           // natural order: lane t gets elements t + T a
        middle_passes<G, 1, 0, TWLDS>(t, x, tw, slab, noclk, twtable);   // its pass-1 store rewrites exactly those slots
        phase_fetch<G, NPASS>(t, x, slab);
        phase_last<G>(x);
        if (active) phase_accumulate(x, acc, P);
    }

    // partial spectrum: only the first N of the M convolution outputs are bins
    exchange_sync<true>();
    double* const stage = reinterpret_cast<double*>(smem);
    constexpr int SM = M + M / 16;
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const int bin = bin_of<G>(t, a);
        stage[fs * SM + bin + (bin >> 4)] = acc[a];
    }
    exchange_sync<true>();
    double* out = partial + static_cast<size_t>(blockIdx.x) * N;
    for (int bin = tid; bin < N; bin += WG) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < FPW; ++k) v += stage[k * SM + bin + (bin >> 4)];
        out[bin] = v;
    }
}

// K3.  out[hop][bin] = (accumulate ? out[hop][bin] : 0) + the partial spectra of the hop's slot
// range, in a fixed order (bit-reproducible for a given grid): thread (g, b) sums the slots
// g, g+GROUPS, g+2 GROUPS, ... of the bin pair b -- 16-byte loads, UNROLL of them in flight -- then
// the GROUPS group sums are added in group order.  blockIdx.y = hop.
// skip (may be null): a device word; non-zero = the partial spectra are not a result (the fused four-step
// kernel gave up) and `out` is left as it is.
template <typename PT, int PAIRS, int GROUPS, int UNROLL>
__global__ __launch_bounds__(PAIRS* GROUPS) void reduce_kernel(
    const PT* __restrict__ partial, const SlotRanges slots, int N, double* __restrict__ out,
    int accumulate, size_t stride, const unsigned* __restrict__ skip)
{
    typedef PT pt2 __attribute__((ext_vector_type(2)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ d2 red[GROUPS][PAIRS + 1];
    if (skip != nullptr && *skip != 0) return;
    const int hop = blockIdx.y;
    const int first = slots.begin[hop], nslots = slots.begin[hop + 1] - first;
    const int b = threadIdx.x % PAIRS, g = threadIdx.x / PAIRS;
    const int bin = blockIdx.x * (2 * PAIRS) + 2 * b;            // N is even: a pair never straddles the end
    d2 s = {0.0, 0.0};
    if (bin < N) {
        const PT* p = partial + static_cast<size_t>(first) * stride + bin;
        int sl = g;
        for (; sl + (UNROLL - 1) * GROUPS < nslots; sl += UNROLL * GROUPS) {
            pt2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = *reinterpret_cast<const pt2*>(p + static_cast<size_t>(sl + u * GROUPS) * stride);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
            }
        }
        for (; sl < nslots; sl += GROUPS) {
            const pt2 v = *reinterpret_cast<const pt2*>(p + static_cast<size_t>(sl) * stride);
            s.x += v.x;
            s.y += v.y;
        }
    }
    red[g][b] = s;
    __syncthreads();
    if (g == 0 && bin < N) {
        double* o = out + static_cast<size_t>(hop) * N + bin;
        d2 tot = {0.0, 0.0};
        if (accumulate) tot = *reinterpret_cast<const d2*>(o);
#pragma unroll
        for (int k = 0; k < GROUPS; ++k) {
            tot.x += red[k][b].x;
            tot.y += red[k][b].y;
        }
        *reinterpret_cast<d2*>(o) = tot;
    }
}

// ---------------------------------------------------------------- dispatch --
using SingleFn = void (*)(const uint8_t*, long, const cf*, const float*, double*);
using ScanFn = void (*)(const cf*, const float*, double*, const HopArgs);

struct Variant {
    int N, vid, P, WG, fpw, lds_bytes;
    bool partial_f32;
    SingleFn single[2][2];   // [window][dma]: one acquisition per launch
    ScanFn scan[2][2];       // several hops per launch
};

// OCC (OCCW for the windowed kernels) = waves per SIMD the register budget
// must admit (= resident workgroups per CU x WG/256).  vid = tuning variant
// (0 = the default for this N).
template <int N, int P, int OCC, int OCCW = OCC, bool DBUF = false, int ACCB = 0, bool PF32 = false,
          int RAWD = 2, int ABL = 0, bool TWLDS = false, int WGO = 0>
Variant make_variant(int vid)
{
    using G = Geom<N, P>;
    constexpr int WG = WGO ? WGO : (G::T >= 256 ? G::T : 256);   // WGO: several frames per workgroup
    constexpr int FPW = WG / G::T;
    constexpr int LDS = FPW * ((DBUF ? 2 : 1) * G::LDS_CPX * (int)sizeof(cf) + RAWD * 2 * N) +
                        (TWLDS ? twlds_entries<G>() * (int)sizeof(cf) : 0);
    return Variant{N, vid, P, WG, FPW, LDS, PF32,
                   {{fft_accum_kernel<G, WG, OCC, false, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>,
                     fft_accum_kernel<G, WG, OCC, false, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>},
                    {fft_accum_kernel<G, WG, OCCW, true, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>,
                     fft_accum_kernel<G, WG, OCCW, true, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>}},
                   {{fft_accum_scan_kernel<G, WG, OCC, false, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>,
                     fft_accum_scan_kernel<G, WG, OCC, false, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>},
                    {fft_accum_scan_kernel<G, WG, OCCW, true, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>,
                     fft_accum_scan_kernel<G, WG, OCCW, true, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS>}}};
}

// the matrix-pipe first pass (N = 4096, rectangular window, LDS-DMA staging only; the other three
// table entries keep the VALU kernel so that a windowed or misaligned launch still runs)
template <int N, int P, int OCC, int WG, int RAWD>
Variant make_mfma_variant(int vid)
{
    Variant v = make_variant<N, P, OCC, OCC, false, 0, false, RAWD, 0, true, WG>(vid);
    v.single[0][1] = fft_accum_mfma_kernel<Geom<N, P>, WG, OCC, RAWD>;
    return v;
}

const Variant kVariants[] = {
    // defaults.  Template arguments after <N, P>: OCC, OCCW, DBUF, ACCB, PF32, RAWD, ABL, TWLDS, WGO
    // 128 = 16 x 8 and 256 = 16 x 16: two passes and ONE exchange at 16 points per lane (measured
    // 12-14 % faster than 8 x 8 x 2 / 8 x 8 x 4 -- the LDS stores are what costs)
    make_variant<64, 8, 4, 4, false, 0, false, 4>(0),    make_variant<128, 16, 3, 3, false, 0, false, 4>(0),
    make_variant<256, 16, 3, 3, false, 0, false, 4>(0),  make_variant<512, 8, 4, 4, false, 0, false, 2>(0),
    make_variant<1024, 16, 3, 2, false, 0, false, 2, 0, true>(0),
    // 2048/4096: one 512-thread workgroup per CU (4 / 2 frames side by side): as fast as three
    // 256-thread workgroups (the kernel is VALU-bound at 8 waves) and a third of the partials.
    make_variant<2048, 16, 2, 2, false, 0, false, 2, 0, true, 512>(0),
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true, 512>(0),
    make_variant<8192, 16, 2, 2, false, 0, false, 2>(0),
#ifdef RPF_TUNING
    // Lab equipment, compiled only into the -DRPF_TUNING build (make tuning ->
    // librpf_engine_tuning.so, used by tools/): in the shipped library every N has exactly
    // one kernel and RPF_FLAG_VARIANT(k != 0) fails rpf_engine_create with
    // RPF_ERR_INVALID_ARGUMENT.  Every variant is exact unless it says float32 or ablation.
    // (Round 2's de-phasing experiments -- alternating frame groups, skewed workgroup starts,
    // register prefetch instead of LDS-DMA -- measured slower and were removed in round 3;
    // their numbers stay in profiles/r02_k1_dephasing.txt and DESIGN.md.)
    make_variant<4096, 16, 3, 2, false, 0, false, 2>(1),              // all twiddles in registers
    make_variant<4096, 16, 3, 2, false, 0, false, 1, 0, true>(2),     // one frame ahead only
    make_variant<4096, 16, 2, 2, true, 0, false, 2, 0, true>(3),      // double-buffered slab (one barrier per frame)
    make_variant<4096, 16, 3, 3, false, 8, true, 2, 0, true>(4),      // float32 batch accumulate + float32 partials
    make_variant<4096, 8, 4, 4, false, 0, false, 2>(5),               // 8 points per lane, 512 threads
    make_variant<4096, 16, 3, 3, false, 0, false, 2, 0, true, 768>(8),   // one 768-thread workgroup per CU, 3 frames side by side
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true>(9),               // 256 threads, 3 (windowed: 2) workgroups per CU
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true>(10),              // 256 threads, 2 workgroups per CU
    make_variant<4096, 16, 2, 2, false, 8, false, 2, 0, true, 512>(22),  // 512 threads, float32 batch accumulate, f64 partials
    make_variant<4096, 16, 2, 2, false, 108, false, 2, 0, true, 512>(23),  // round 6: PACKED float32 pre-accumulate over 8 frames
    make_variant<4096, 16, 2, 2, false, 116, false, 2, 0, true, 512>(24),  // ... over 16 frames
    make_variant<4096, 16, 2, 2, true, 0, false, 1, 0, true, 512>(26),   // 512 threads, double-buffered slab (no top barrier), raw ring 1
    // round 3: deeper raw rings for HBM-resident input (one workgroup per CU leaves the LDS for it),
    // pass-2 twiddles back in registers (the 512-thread form has the registers)
    make_variant<4096, 16, 2, 2, false, 0, false, 3, 0, true, 512>(50),
    make_variant<4096, 16, 2, 2, false, 0, false, 4, 0, true, 512>(51),
    make_variant<4096, 16, 2, 2, false, 0, false, 6, 0, true, 512>(52),
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, false, 512>(53),
    make_variant<4096, 16, 2, 2, false, 0, false, 4, 0, false, 512>(54),
    make_variant<2048, 16, 2, 2, false, 0, false, 4, 0, true, 512>(51),
    make_mfma_variant<4096, 16, 2, 512, 2>(60),              // first pass on the matrix pipe (exact, slower: see the kernel)
    make_variant<512, 8, 4, 4, false, 0, false, 4>(1),  make_variant<512, 8, 4, 4, false, 0, false, 8>(2),
    make_variant<512, 16, 3, 3, false, 0, false, 2>(3),
    make_variant<128, 8, 4, 4, false, 0, false, 4>(3),   make_variant<256, 8, 4, 4, false, 0, false, 4>(3),    // 8 points per lane, three passes
    make_variant<1024, 8, 4, 4, false, 0, false, 4>(1), make_variant<2048, 8, 4, 4, false, 0, false, 4>(1),
    make_variant<1024, 16, 3, 3, false, 0, false, 2>(2), make_variant<2048, 16, 3, 3, false, 0, false, 2>(2),
    make_variant<1024, 16, 2, 2, false, 0, false, 2, 0, true, 512>(9),
    make_variant<2048, 16, 3, 2, false, 0, false, 2, 0, true>(9),
    make_variant<512, 8, 2, 2, false, 0, false, 2, 0, false, 512>(9),
    make_variant<8192, 16, 2, 2, false, 0, false, 2, 0, true>(1),
    // measurement-only ablations of the default N=4096 kernel (results are garbage)
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 1, true, 512>(31),    // no accumulate
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 2, true, 512>(32),    // no butterfly arithmetic
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 4, true, 512>(34),    // no LDS exchanges
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 8, true, 512>(38),    // no HBM staging
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 13, true, 512>(36),   // butterflies only
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 12, true, 512>(37),   // butterflies + accumulate only
#endif  // RPF_TUNING
};

const Variant* find_variant(int N, int vid)
{
    for (const Variant& v : kVariants)
        if (v.N == N && v.vid == vid) return &v;
    return nullptr;
}

using BluesteinFn = void (*)(const uint8_t*, long, int, const cf*, const cf*, const cf*, double*);
struct BluesteinVariant {
    int M, WG, fpw, lds_bytes;
    BluesteinFn fn;
};
template <int M, int P, int OCC, bool TWLDS = false>
BluesteinVariant make_bluestein()
{
    using G = Geom<M, P>;
    constexpr int WG = G::T >= 256 ? G::T : 256;
    constexpr int FPW = WG / G::T;
    constexpr int LDS = FPW * G::LDS_CPX * (int)sizeof(cf) + (TWLDS ? twlds_entries<G>() * (int)sizeof(cf) : 0);
    return BluesteinVariant{M, WG, FPW, LDS, bluestein_kernel<G, WG, OCC, TWLDS>};
}
const BluesteinVariant kBluestein[] = {
    make_bluestein<64, 8, 4>(),    make_bluestein<128, 8, 4>(),   make_bluestein<256, 8, 4>(),
    // P = 8 up to M = 1024; from M = 2048 on, 16 points per lane with the pass-2/3
    // twiddles in an LDS table (one pass and one exchange less per transform; with
    // register twiddles two inlined 16-point transforms spill even at 256 VGPRs).
    // Measured: M = 4096 +24 %, 2048 +6 %, 1024 +-0, 512 -6 %.
    make_bluestein<512, 8, 4>(),   make_bluestein<1024, 8, 2>(),  make_bluestein<2048, 16, 2, true>(),
    make_bluestein<4096, 16, 2, true>(),  make_bluestein<8192, 16, 2, true>(),
};
const BluesteinVariant* find_bluestein(int M)
{
    for (const BluesteinVariant& v : kBluestein)
        if (v.M == M) return &v;
    return nullptr;
}

}  // namespace

bool kernel_supported(int N, int vid) { return find_variant(N, vid) != nullptr; }

#ifdef RPF_PHASE_TIMING
extern "C" int rpf_debug_phase_cycles(unsigned long long* out16, unsigned long long* waves, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * kPhaseSlots) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(waves, HIP_SYMBOL(g_phase_waves), sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[kPhaseSlots] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_waves), z, sizeof(unsigned long long));
    }
    return 0;
}
#endif

hipError_t plan_launch(int N, int vid, bool window, bool use_dma, int device, LaunchInfo* li)
{
    const Variant* v = find_variant(N, vid);
    if (!v) return hipErrorInvalidValue;
    int per_cu = 1 << 30;
    for (int scan = 0; scan < 2; ++scan) {          // the single-acquisition and the scan instantiation share one grid
        const void* fn = scan ? reinterpret_cast<const void*>(v->scan[window ? 1 : 0][use_dma ? 1 : 0])
                              : reinterpret_cast<const void*>(v->single[window ? 1 : 0][use_dma ? 1 : 0]);
        hipError_t err = hipFuncSetAttribute(fn,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes);
        if (err != hipSuccess) return err;
        int n = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, v->WG, v->lds_bytes);
        if (err != hipSuccess) return err;
        per_cu = std::min(per_cu, n);
    }
    hipDeviceProp_t prop;
    hipError_t err = hipGetDeviceProperties(&prop, device);
    if (err != hipSuccess) return err;
    if (per_cu < 1) per_cu = 1;
    li->grid = per_cu * prop.multiProcessorCount;
    li->block = v->WG;
    li->fpw = v->fpw;
    li->lds_bytes = v->lds_bytes;
    li->partial_f32 = v->partial_f32;
    return hipSuccess;
}

hipError_t launch_fft_accum(int N, int vid, bool window, bool use_dma, const uint8_t* d_stream,
                            long nframes, const cf* d_twiddles, const float* d_window,
                            double* d_partial, int grid, hipStream_t stream, LaunchInfo* li)
{
    const Variant* v = find_variant(N, vid);
    if (!v || grid < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(v->single[window ? 1 : 0][use_dma ? 1 : 0], dim3(grid), dim3(v->WG), v->lds_bytes, stream,
                       d_stream, nframes, d_twiddles, d_window, d_partial);
    if (li) {
        li->grid = grid;
        li->block = v->WG;
        li->fpw = v->fpw;
        li->lds_bytes = v->lds_bytes;
        li->partial_f32 = v->partial_f32;
    }
    return hipGetLastError();
}

hipError_t launch_fft_accum_hops(int N, int vid, bool window, bool use_dma, const HopArgs& hops,
                                 const cf* d_twiddles, const float* d_window, double* d_partial, int grid,
                                 hipStream_t stream, LaunchInfo* li)
{
    const Variant* v = find_variant(N, vid);
    if (!v || grid < 1 || hops.H < 1 || hops.H > kMaxHops || grid > hops.total) return hipErrorInvalidValue;
    hipLaunchKernelGGL(v->scan[window ? 1 : 0][use_dma ? 1 : 0], dim3(grid), dim3(v->WG), v->lds_bytes, stream,
                       d_twiddles, d_window, d_partial, hops);
    if (li) {
        li->grid = grid;
        li->block = v->WG;
        li->fpw = v->fpw;
        li->lds_bytes = v->lds_bytes;
        li->partial_f32 = v->partial_f32;
    }
    return hipGetLastError();
}

bool bluestein_supported(int N)
{
    return N >= 2 && N % 2 == 0 && N <= 4096 && !kernel_supported(N, 0) &&
           find_bluestein(bluestein_length(N)) != nullptr;
}

hipError_t plan_bluestein(int N, int device, LaunchInfo* li)
{
    const BluesteinVariant* v = bluestein_supported(N) ? find_bluestein(bluestein_length(N)) : nullptr;
    if (!v) return hipErrorInvalidValue;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(v->fn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes);
    if (err != hipSuccess) return err;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(v->fn),
                                                       v->WG, v->lds_bytes);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    err = hipGetDeviceProperties(&prop, device);
    if (err != hipSuccess) return err;
    li->grid = std::max(per_cu, 1) * prop.multiProcessorCount;
    li->block = v->WG;
    li->fpw = v->fpw;
    li->lds_bytes = v->lds_bytes;
    return hipSuccess;
}

hipError_t launch_bluestein(int N, const uint8_t* d_stream, long nframes, const cf* d_twM,
                            const cf* d_g, const cf* d_bhat, double* d_partial, int grid,
                            hipStream_t stream, LaunchInfo* li)
{
    const BluesteinVariant* v = bluestein_supported(N) ? find_bluestein(bluestein_length(N)) : nullptr;
    if (!v || grid < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(v->fn, dim3(grid), dim3(v->WG), v->lds_bytes, stream, d_stream, nframes, N, d_twM,
                       d_g, d_bhat, d_partial);
    if (li) {
        li->grid = grid;
        li->block = v->WG;
        li->fpw = v->fpw;
        li->lds_bytes = v->lds_bytes;
    }
    return hipGetLastError();
}

namespace {
template <typename PT, int PAIRS, int GROUPS, int UNROLL>
void launch_reduce_shape(const PT* d_partial, const SlotRanges& slots, int H, int N, double* d_out, bool accumulate,
                         hipStream_t stream, size_t stride, const unsigned* d_skip = nullptr)
{
    const dim3 blocks((N + 2 * PAIRS - 1) / (2 * PAIRS), H);
    hipLaunchKernelGGL((reduce_kernel<PT, PAIRS, GROUPS, UNROLL>), blocks, dim3(PAIRS * GROUPS), 0, stream, d_partial,
                       slots, N, d_out, accumulate ? 1 : 0, stride, d_skip);
}
}  // namespace

hipError_t launch_reduce_hops(const double* d_partial, const SlotRanges& slots, int H, int N, double* d_out,
                              bool accumulate, hipStream_t stream, bool partial_f32, size_t slot_stride,
                              const unsigned* d_skip)
{
    if (H < 1 || H > kMaxHops) return hipErrorInvalidValue;
    const size_t stride = slot_stride ? slot_stride : static_cast<size_t>(N);
    if (partial_f32) {
        launch_reduce_shape<float, 8, 32, 8>(reinterpret_cast<const float*>(d_partial), slots, H, N, d_out, accumulate,
                                             stream, stride, d_skip);
        return hipGetLastError();
    }
    // one block shape: seven were measured in situ behind K1 within 0.3 us of each other (profiles/r03_k3_shapes.txt)
    launch_reduce_shape<double, 8, 16, 8>(d_partial, slots, H, N, d_out, accumulate, stream, stride, d_skip);
    return hipGetLastError();
}

hipError_t launch_reduce(const double* d_partial, int nslots, int N, double* d_out,
                         bool accumulate, hipStream_t stream, bool partial_f32, size_t slot_stride,
                         const unsigned* d_skip)
{
    SlotRanges one;
    one.begin[0] = 0;
    for (int h = 1; h <= kMaxHops; ++h) one.begin[h] = nslots;
    return launch_reduce_hops(d_partial, one, 1, N, d_out, accumulate, stream, partial_f32, slot_stride, d_skip);
}

void make_twiddles(int N, std::vector<cf>& out)
{
    out.resize(N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < N; ++k) {
        const long double a = two_pi * static_cast<long double>(k) / static_cast<long double>(N);
        out[k].x = static_cast<float>(cosl(a));
        out[k].y = static_cast<float>(-sinl(a));
    }
}

}  // namespace rpf
