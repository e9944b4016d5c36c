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
#include <type_traits>

#include <algorithm>

#include "bluestein_tables.h"
#include "rpf_device_common.h"
#include "rpf_kernels.h"

namespace rpf {

namespace {

// One 16-byte piece of the frame data a lane stages (same mapping as stage_raw),
// loaded into registers.
template <class G>
__device__ __forceinline__ uint4 load_raw_piece(const uint8_t* __restrict__ stream, long fb, long nframes,
                                                int wave, int lane, int i)
{
    int slot, off;
    raw_source<G>(wave, i * 1024 + lane * 16, &slot, &off);
    long f = fb + slot;
    f = f < nframes ? f : nframes - 1;
    return *reinterpret_cast<const uint4*>(stream + f * (2 * G::N) + off);
}

// Stage the raw bytes this wavefront will unpack in the iteration whose slot-0
// frame is `fb` (wave-local, a-major layout: fft_core.h raw_source).  P/8
// instructions per wave, each moving 64 lanes x 16 B = eight 128-byte rows.
// Frames past the end are clamped to the last frame (never accumulated) so that
// every iteration issues the same number of DMA instructions and the counted
// s_waitcnt vmcnt(N) at the top of the frame loop stays exact.
template <class G, bool DMA>
__device__ __forceinline__ void stage_raw(const uint8_t* __restrict__ stream, long fb,
                                          long nframes, uint8_t* wave_raw, int wave, int lane)
{
    constexpr int PIECES = G::P / 8;
    constexpr int FRAME_BYTES = 2 * G::N;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int j = i * 1024 + lane * 16;
        int slot, off;
        raw_source<G>(wave, j, &slot, &off);
        long f = fb + slot;
        f = f < nframes ? f : nframes - 1;
        const uint8_t* src = stream + f * FRAME_BYTES + off;
        if constexpr (DMA) {
            // LDS address = wave-uniform base + 16 * lane (added by the hardware)
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wave_raw + i * 1024), 16, 0, 0);
        } else {
            *reinterpret_cast<uint4*>(wave_raw + j) = *reinterpret_cast<const uint4*>(src);
        }
    }
}

// DBUF: two slabs used alternately.  With one slab the pass-1 store of frame f+1
// must wait (workgroup barrier at the top of the loop) until every wave has
// finished reading frame f's slab; with two, the single barrier after the pass-1
// store orders both hazards and a frame costs one s_barrier instead of two, at
// the price of LDS (fewer resident workgroups).
//
// RAWD: depth of the raw-byte ring = frames staged ahead by LDS-DMA.  The HBM
// latency seen by a DMA under load is several frame times (measured ~3 us vs
// ~1 us of butterflies per frame), so one frame ahead leaves the workgroup idle
// most of the time; RAWD frames ahead keep RAWD x 2N bytes per workgroup in flight.
// ACCB > 0 (tuning variants): |X|^2 is first summed over ACCB frames in packed
// float32 (one v_pk_fma_f32 per bin instead of four half-rate f64 instructions)
// and only then folded into the f64 accumulators -- adds <= ~1e-7 relative error
// per batch, averaged down over the batches.  PF32: partial spectra leave as
// float32 (half the flush and K3 traffic; each partial is a sum over ~13 frames
// and there are hundreds of them, so the rounding averages out to ~1e-9).
__device__ unsigned int g_cu_tickets[1024];   // SKEW variants: arrival counters per CU (never reset: used modulo)

// SKEW (tuning): workgroups that share a CU start SKEW x 64 cycles apart (by the slot their
// waves got on the SIMD), so that their VALU and LDS phases interleave instead of colliding.
template <class G, int WG, int OCC, bool WINDOW, bool DMA, bool DBUF, int ACCB = 0, bool PF32 = false,
          int RAWD = 2, int ABL = 0, bool TWLDS = false, bool RAWREG = false, int SKEW = 0>
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
    uint4 pre[RAWREG ? PIECES : 1];
    if (fb < nframes) {
        if constexpr (RAWREG) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) pre[i] = load_raw_piece<G>(stream, fb, nframes, wave, lane, i);
        } else {
#pragma unroll
            for (int d = 0; d < RAWD; ++d)
                stage_raw<G, DMA>(stream, fb + d * stride, nframes, wave_raw + d * RAW_SLOT, wave, lane);
        }
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
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = 0.0;
    if constexpr (ACCB > 0) {
#pragma unroll
        for (int a = 0; a < P; ++a) acc32[a] = 0.0f;
    }

    if constexpr (SKEW > 0) {
        // the k-th workgroup to arrive on this CU (ticket from a per-CU counter keyed by
        // HW_REG_XCC_ID and HW_REG_HW_ID's se/sh/cu fields) starts k x SKEW cycles late
        constexpr int PER_CU = OCC * 256 / WG;
        int* const box = reinterpret_cast<int*>(smem);          // the slab is not in use yet
        if (tid == 0) {
            const unsigned cu = __builtin_amdgcn_s_getreg((6 << 11) | (8 << 6) | 4);     // HW_ID[14:8]
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID[3:0]
            box[0] = static_cast<int>(atomicAdd(&g_cu_tickets[((xcc << 7) | cu) & 1023u], 1u) % PER_CU);
        }
        exchange_sync<true>();
        const int ticket = box[0];
        exchange_sync<true>();
#pragma unroll 1
        for (int k = 0; k < ticket * (SKEW / 512); ++k) __builtin_amdgcn_s_sleep(8);
    }
    PhaseClock clk;
    clk.start();
    for (int it = 0; fb < nframes; fb += stride, ++it) {
        const bool active = (fb + fs) < nframes;
        cf* const slab = slab_base + ((DBUF ? (it & 1) : 0) * FPW + fs) * G::LDS_CPX;
        uint8_t* const ring_slot = wave_raw + (it % RAWD) * RAW_SLOT;
        cf x[P];

        if constexpr (RAWREG) {
            // this frame's bytes sit in VGPRs: drop them into the wave's raw slot, then
            // start the loads of the next frame into the same registers
#pragma unroll
            for (int i = 0; i < PIECES; ++i) *reinterpret_cast<uint4*>(ring_slot + i * 1024 + lane * 16) = pre[i];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) pre[i] = load_raw_piece<G>(stream, fb + stride, nframes, wave, lane, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 0);
            phase_unpack<G, WINDOW>(ring_slot + 2 * lane, sgn, wsgn, x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            exchange_sync<false>();
            RPF_STAMP(clk, 1);
        } else {
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
                stage_raw<G, DMA>(stream, fb + RAWD * stride, nframes, ring_slot, wave, lane);
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
    clk.publish(lane);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) prefetches
    if constexpr (ACCB > 0) {
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

// K1, alternating form (N = 2048 / 4096: several frames per 512-thread workgroup, a frame spans
// more than one wavefront).  In fft_accum_kernel the frames of a workgroup run in lock-step: every
// wave is in its butterflies at the same time (the LDS idles) and in its exchange at the same time
// (the VALUs idle) -- measured, a frame round costs the SUM of its VALU and LDS phases.  Here the
// frame slots form two groups that run half an iteration apart, swapping roles at each of the two
// workgroup barriers a frame needs anyway:
//     half-step h:   group 0: phase (h & 1) of frame h / 2,   group 1: phase ((h - 1) & 1) of frame (h - 1) / 2
//     phase 0: [|X|^2 of the previous frame] unpack, pass-1 butterflies, pass-1 store
//     phase 1: pass-2 fetch ... last pass
// so that on every SIMD one wave's arithmetic runs beside the other wave's LDS traffic and
// barrier wait.  Same arithmetic per frame as fft_accum_kernel: results are bit-identical.
template <class G, int WG, int OCC, bool WINDOW, bool DMA, int RAWD = 2, bool TWLDS = true, bool ACC_LATE = true>
__global__ __launch_bounds__(WG, OCC) void fft_accum_alt_kernel(const uint8_t* __restrict__ stream,
                                                                long nframes,
                                                                const cf* __restrict__ twN,
                                                                const float* __restrict__ window,
                                                                double* __restrict__ partial)
{
    constexpr int P = G::P, T = G::T, N = G::N, NPASS = G::NPASS;
    constexpr int FPW = WG / T;
    static_assert(WG % T == 0 && T % 64 == 0 && FPW % 2 == 0, "two groups of whole wavefronts");
    static_assert(NPASS >= 2 && (NPASS == 2 || G::Lcur(2) <= 64), "only the first exchange crosses wavefronts");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* const slab_base = reinterpret_cast<cf*>(smem);                   // [FPW][LDS_CPX]
    uint8_t* const raw_base = smem + FPW * G::LDS_CPX * sizeof(cf);      // [WG/64][RAWD][128 P]

    const int tid = threadIdx.x;
    const int fs = tid / T, t = tid % T;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int grp = ((wave * 64) / T) & 1;                               // wave-uniform
    constexpr int RAW_SLOT = kRawChunk * P;
    constexpr int PIECES = P / 8;
    uint8_t* const wave_raw = raw_base + wave * (RAWD * RAW_SLOT);
    cf* const slab = slab_base + fs * G::LDS_CPX;

    const long stride = static_cast<long>(gridDim.x) * FPW;
    const long fb0 = static_cast<long>(blockIdx.x) * FPW;
    const int iters = fb0 < nframes ? static_cast<int>((nframes - fb0 + stride - 1) / stride) : 0;
    if (iters > 0) {
#pragma unroll
        for (int d = 0; d < RAWD; ++d)
            stage_raw<G, DMA>(stream, fb0 + d * stride, nframes, wave_raw + d * RAW_SLOT, wave, lane);
    }

    cf tw[NPASS - 1][P - 1];
    load_twiddles<G, 1, TWLDS>(t, twN, tw);
    cf* const twtable = reinterpret_cast<cf*>(raw_base + (WG / 64) * RAWD * RAW_SLOT);
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
#pragma unroll
    for (int a = 0; a < P; ++a) acc[a] = 0.0;

    PhaseClock clk;
    cf x[P];
    bool pending = false;        // x holds the spectrum of a frame that has not been accumulated yet
#pragma unroll 1
    for (int h = 0; h <= 2 * iters; ++h) {
        const int hh = h - grp;
        const int it = hh >> 1;
        if (hh >= 0 && it < iters) {
            const long fb = fb0 + it * stride;
            if ((hh & 1) == 0) {
                if constexpr (ACC_LATE) {
                    if (pending) phase_accumulate(x, acc, P);
                }
                uint8_t* const ring_slot = wave_raw + (it % RAWD) * RAW_SLOT;
                if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RAWD - 1) * PIECES) : "memory");
                exchange_sync<false>();
                phase_unpack<G, WINDOW>(ring_slot + 2 * lane, sgn, wsgn, x);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot's reads have returned
                exchange_sync<false>();
                stage_raw<G, DMA>(stream, fb + RAWD * stride, nframes, ring_slot, wave, lane);
                phase_butterfly_twiddle<G>(x, tw[0]);
                phase_store<G, 1>(t, x, slab);
            } else {
                if constexpr (NPASS > 2) middle_passes<G, 2, 0, TWLDS>(t, x, tw, slab, clk, twtable);
                phase_fetch<G, NPASS>(t, x, slab);
                phase_last<G>(x);
                const bool active = (fb + fs) < nframes;
                if constexpr (ACC_LATE) {
                    pending = active;
                } else {
                    if (active) phase_accumulate(x, acc, P);
                }
            }
        }
        exchange_sync<true>();       // the two groups swap roles
    }
    if constexpr (ACC_LATE) {
        if (pending) phase_accumulate(x, acc, P);
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (clamped) prefetches

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
    for (int bin = tid; bin < N; bin += WG) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < FPW; ++k) v += stage[k * SN + bin + (bin >> 4)];
        partial[static_cast<size_t>(blockIdx.x) * N + bin] = v;
    }
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

// K3.  out[bin] = (accumulate ? out[bin] : 0) + sum over workgroup partials, in
// a fixed order (bit-reproducible for a given grid): thread (g, b) sums the
// partials g, g+16, g+32, ... of bin b with 8 independent loads in flight, then
// the 16 group sums are added in group order.
constexpr int RED_BINS = 16, RED_GROUPS = 16, RED_UNROLL = 8;

template <typename PT>
__global__ __launch_bounds__(RED_BINS* RED_GROUPS) void reduce_kernel(
    const PT* __restrict__ partial, int nslots, int N, double* __restrict__ out,
    int accumulate, size_t stride)
{
    __shared__ double red[RED_GROUPS][RED_BINS + 1];
    const int b = threadIdx.x % RED_BINS, g = threadIdx.x / RED_BINS;
    const int bin = blockIdx.x * RED_BINS + b;
    double s = 0.0;
    if (bin < N) {
        const PT* p = partial + bin;
        int sl = g;
        for (; sl + (RED_UNROLL - 1) * RED_GROUPS < nslots; sl += RED_UNROLL * RED_GROUPS) {
            PT v[RED_UNROLL];
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u)
                v[u] = p[static_cast<size_t>(sl + u * RED_GROUPS) * stride];
#pragma unroll
            for (int u = 0; u < RED_UNROLL; ++u) s += v[u];
        }
        for (; sl < nslots; sl += RED_GROUPS) s += p[static_cast<size_t>(sl) * stride];
    }
    red[g][b] = s;
    __syncthreads();
    if (g == 0 && bin < N) {
        double tot = accumulate ? out[bin] : 0.0;
#pragma unroll
        for (int k = 0; k < RED_GROUPS; ++k) tot += red[k][b];
        out[bin] = tot;
    }
}

// ---------------------------------------------------------------- dispatch --
using KernelFn = void (*)(const uint8_t*, long, const cf*, const float*, double*);

struct Variant {
    int N, vid, P, WG, fpw, lds_bytes;
    bool partial_f32;
    KernelFn fn[2][2];   // [window][dma]
};

// OCC (OCCW for the windowed kernels) = waves per SIMD the register budget
// must admit (= resident workgroups per CU x WG/256).  vid = tuning variant
// (0 = the default for this N).
template <int N, int P, int OCC, int OCCW = OCC, bool DBUF = false, int ACCB = 0, bool PF32 = false,
          int RAWD = 2, int ABL = 0, bool TWLDS = false, bool RAWREG = false, int WGO = 0, int SKEW = 0>
Variant make_variant(int vid)
{
    using G = Geom<N, P>;
    constexpr int WG = WGO ? WGO : (G::T >= 256 ? G::T : 256);   // WGO: several frames per workgroup
    constexpr int FPW = WG / G::T;
    constexpr int LDS = FPW * ((DBUF ? 2 : 1) * G::LDS_CPX * (int)sizeof(cf) + RAWD * 2 * N) +
                        (TWLDS ? twlds_entries<G>() * (int)sizeof(cf) : 0);
    return Variant{N, vid, P, WG, FPW, LDS, PF32,
                   {{fft_accum_kernel<G, WG, OCC, false, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS, RAWREG, SKEW>,
                     fft_accum_kernel<G, WG, OCC, false, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS, RAWREG, SKEW>},
                    {fft_accum_kernel<G, WG, OCCW, true, false, DBUF, ACCB, PF32, RAWD, ABL, TWLDS, RAWREG, SKEW>,
                     fft_accum_kernel<G, WG, OCCW, true, true, DBUF, ACCB, PF32, RAWD, ABL, TWLDS, RAWREG, SKEW>}}};
}

// the alternating form: WG threads = WG / T frame slots in two groups half an iteration apart
template <int N, int P, int WG, int OCC, int RAWD = 2, bool TWLDS = true, bool ACC_LATE = true>
Variant make_alt_variant(int vid)
{
    using G = Geom<N, P>;
    constexpr int FPW = WG / G::T;
    constexpr int LDS = FPW * (G::LDS_CPX * (int)sizeof(cf) + RAWD * 2 * N) +
                        (TWLDS ? twlds_entries<G>() * (int)sizeof(cf) : 0);
    return Variant{N, vid, P, WG, FPW, LDS, false,
                   {{fft_accum_alt_kernel<G, WG, OCC, false, false, RAWD, TWLDS, ACC_LATE>,
                     fft_accum_alt_kernel<G, WG, OCC, false, true, RAWD, TWLDS, ACC_LATE>},
                    {fft_accum_alt_kernel<G, WG, OCC, true, false, RAWD, TWLDS, ACC_LATE>,
                     fft_accum_alt_kernel<G, WG, OCC, true, true, RAWD, TWLDS, ACC_LATE>}}};
}

const Variant kVariants[] = {
    // defaults.  Template arguments after <N, P>: OCC, OCCW, DBUF, ACCB, PF32, RAWD, ABL, TWLDS
    // 128 = 16 x 8 and 256 = 16 x 16: two passes and ONE exchange at 16 points per lane (measured
    // 12-14 % faster than 8 x 8 x 2 / 8 x 8 x 4 -- the LDS stores are what costs)
    make_variant<64, 8, 4, 4, false, 0, false, 4>(0),    make_variant<128, 16, 3, 3, false, 0, false, 4>(0),
    make_variant<256, 16, 3, 3, false, 0, false, 4>(0),  make_variant<512, 8, 4, 4, false, 0, false, 2>(0),
    make_variant<1024, 16, 3, 2, false, 0, false, 2, 0, true>(0),
    // 2048/4096: one 512-thread workgroup per CU (4 / 2 frames side by side): as fast as three
    // 256-thread workgroups (the kernel is VALU-bound at 8 waves) and a third of the partials.
    make_variant<2048, 16, 2, 2, false, 0, false, 2, 0, true, false, 512>(0),
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true, false, 512>(0),
    make_variant<8192, 16, 2, 2, false, 0, false, 2>(0),
#ifdef RPF_TUNING
    // Lab equipment, compiled only into the -DRPF_TUNING build (make tuning ->
    // librpf_engine_tuning.so, used by tools/): in the shipped library every N has exactly
    // one kernel and RPF_FLAG_VARIANT(k != 0) fails rpf_engine_create with
    // RPF_ERR_INVALID_ARGUMENT.  Every variant is exact unless it says float32 or ablation.
    make_variant<4096, 16, 3, 2, false, 0, false, 2>(1),              // all twiddles in registers
    make_variant<4096, 16, 3, 2, false, 0, false, 1, 0, true>(2),     // one frame ahead only
    make_variant<4096, 16, 2, 2, true, 0, false, 2, 0, true>(3),      // double-buffered slab (one barrier per frame)
    make_variant<4096, 16, 3, 3, false, 8, true, 2, 0, true>(4),      // float32 batch accumulate + float32 partials
    make_variant<4096, 8, 4, 4, false, 0, false, 2>(5),               // 8 points per lane, 512 threads
    make_variant<4096, 16, 3, 3, false, 0, false, 2, 0, true, false, 768>(8),   // one 768-thread workgroup per CU, 3 frames side by side
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true>(9),               // 256 threads, 3 (windowed: 2) workgroups per CU
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true>(10),              // 256 threads, 2 workgroups per CU
    make_variant<4096, 16, 2, 2, false, 8, false, 2, 0, true, false, 512>(22),  // 512 threads, float32 batch accumulate, f64 partials
    make_variant<4096, 16, 2, 2, true, 0, false, 1, 0, true, false, 512>(26),   // 512 threads, double-buffered slab (no top barrier), raw ring 1
    make_variant<4096, 16, 3, 2, false, 0, false, 1, 0, true, true>(6),   // next frame prefetched in VGPRs, no LDS-DMA
    make_variant<4096, 16, 3, 3, false, 8, true, 1, 0, true, true>(7),    // same + float32 batch accumulate/partials
    make_variant<512, 8, 4, 4, false, 0, false, 4>(1),  make_variant<512, 8, 4, 4, false, 0, false, 8>(2),
    make_variant<512, 16, 3, 3, false, 0, false, 2>(3),
    make_variant<128, 8, 4, 4, false, 0, false, 4>(3),   make_variant<256, 8, 4, 4, false, 0, false, 4>(3),    // 8 points per lane, three passes
    make_variant<1024, 8, 4, 4, false, 0, false, 4>(1), make_variant<2048, 8, 4, 4, false, 0, false, 4>(1),
    make_variant<1024, 16, 3, 3, false, 0, false, 2>(2), make_variant<2048, 16, 3, 3, false, 0, false, 2>(2),
    make_variant<1024, 16, 2, 2, false, 0, false, 2, 0, true, false, 512>(9),
    make_variant<2048, 16, 3, 2, false, 0, false, 2, 0, true>(9),
    make_variant<512, 8, 2, 2, false, 0, false, 2, 0, false, false, 512>(9),
    make_variant<8192, 16, 2, 2, false, 0, false, 2, 0, true>(1),
    // round 2: de-phasing experiments
    make_alt_variant<4096, 16, 512, 2>(40),                    // alternating groups, |X|^2 deferred into phase 0
    make_alt_variant<4096, 16, 512, 2, 2, true, false>(41),    // alternating groups, |X|^2 at the end of phase 1
    make_alt_variant<4096, 16, 512, 2, 1>(42),                 // ... raw ring of one frame
    make_alt_variant<2048, 16, 512, 2>(40),
    make_alt_variant<2048, 16, 512, 2, 2, true, false>(41),
    make_alt_variant<1024, 16, 512, 2>(40),                    // (T = 64: one wavefront per frame, 8 slots)
    // independent 256-thread workgroups, the k-th on a CU started k x SKEW cycles late
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true, false, 0, 1536>(43),
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true, false, 0, 2560>(44),
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true, false, 0, 3584>(45),
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true, false, 0, 2560>(46),
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 0, true, false, 0, 4096>(47),
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 0, true, false, 0, 512>(48),
    make_variant<4096, 16, 3, 2, false, 0, false, 1, 0, true, false, 0, 2560>(49),     // raw ring of one frame
    // measurement-only ablations of the default N=4096 kernel (results are garbage)
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 1, true>(11),    // no accumulate
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 2, true>(12),    // no butterfly arithmetic
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 4, true>(13),    // no LDS exchanges
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 8, true>(14),    // no HBM staging
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 6, true>(15),    // no arithmetic, no exchanges
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 3, true>(16),    // no arithmetic at all
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 12, true>(17),   // butterflies + accumulate only (no exchanges, no staging)
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 13, true>(18),   // butterflies only
    make_variant<4096, 16, 3, 2, false, 0, false, 2, 14, true>(19),   // accumulate only (+ barriers, unpack)
    // the same ablations on the default 512-thread configuration
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 1, true, false, 512>(31),    // no accumulate
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 2, true, false, 512>(32),    // no butterfly arithmetic
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 4, true, false, 512>(34),    // no LDS exchanges
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 8, true, false, 512>(38),    // no HBM staging
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 13, true, false, 512>(36),   // butterflies only
    make_variant<4096, 16, 2, 2, false, 0, false, 2, 12, true, false, 512>(37),   // butterflies + accumulate only
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
    KernelFn fn = v->fn[window ? 1 : 0][use_dma ? 1 : 0];
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, v->lds_bytes);
    if (err != hipSuccess) return err;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn),
                                                       v->WG, v->lds_bytes);
    if (err != hipSuccess) return err;
    hipDeviceProp_t prop;
    err = hipGetDeviceProperties(&prop, device);
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
    KernelFn fn = v->fn[window ? 1 : 0][use_dma ? 1 : 0];
    hipLaunchKernelGGL(fn, dim3(grid), dim3(v->WG), v->lds_bytes, stream, d_stream, nframes,
                       d_twiddles, d_window, d_partial);
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

hipError_t launch_reduce(const double* d_partial, int nslots, int N, double* d_out,
                         bool accumulate, hipStream_t stream, bool partial_f32, size_t slot_stride)
{
    const int blocks = (N + RED_BINS - 1) / RED_BINS;
    const size_t stride = slot_stride ? slot_stride : static_cast<size_t>(N);
    if (partial_f32)
        hipLaunchKernelGGL(reduce_kernel<float>, dim3(blocks), dim3(RED_BINS * RED_GROUPS), 0, stream,
                           reinterpret_cast<const float*>(d_partial), nslots, N, d_out, accumulate ? 1 : 0, stride);
    else
        hipLaunchKernelGGL(reduce_kernel<double>, dim3(blocks), dim3(RED_BINS * RED_GROUPS), 0, stream,
                           d_partial, nslots, N, d_out, accumulate ? 1 : 0, stride);
    return hipGetLastError();
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
