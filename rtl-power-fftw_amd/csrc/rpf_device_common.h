// rpf_device_common.h -- device-side helpers shared by the gfx950 kernels
// (rpf_kernels.hip: LDS-resident FFT sizes; rpf_fourstep.hip: four-step sizes).
#pragma once

#include <hip/hip_runtime.h>

#include "fft_core.h"

namespace rpf {
namespace {

// Optional per-phase cycle stamps (build with -DRPF_PHASE_TIMING; never in the
// shipped library): each wave adds s_memtime deltas into g_phase_cycles so that
// tools/gpu_phases.py can print where a frame's cycles go.
#ifdef RPF_PHASE_TIMING
constexpr int kPhaseSlots = 16;
__device__ unsigned long long g_phase_cycles[kPhaseSlots];
__device__ unsigned long long g_phase_waves;
struct PhaseClock {
    unsigned long long last;
    unsigned long long sum[kPhaseSlots];
    __device__ __forceinline__ void start()
    {
        for (int i = 0; i < kPhaseSlots; ++i) sum[i] = 0;
        last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void stamp(int i)
    {
        const unsigned long long now = __builtin_readcyclecounter();
        sum[i] += now - last;
        last = now;
    }
    __device__ __forceinline__ void publish(int lane)
    {
        if (lane == 0) {
            for (int i = 0; i < kPhaseSlots; ++i) atomicAdd(&g_phase_cycles[i], sum[i]);
            atomicAdd(&g_phase_waves, 1ull);
        }
    }
};
#define RPF_STAMP(clk, i) (clk).stamp(i)
#else
struct PhaseClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void publish(int) {}
};
#define RPF_STAMP(clk, i) ((void)0)
#endif

// Orders LDS traffic between the threads that exchange data.  A frame that spans
// several wavefronts needs s_barrier; it is issued raw, after waiting for this
// wave's LDS operations only (lgkmcnt): __syncthreads() would also drain vmcnt
// and with it the LDS-DMA prefetch of the coming frames, exposing the full HBM
// latency once per frame (measured: 3/4 of the kernel time).  LDS-DMA data is
// ordered by the issuing wave's own counted vmcnt wait, and a wave only ever
// reads bytes it staged itself.  Exchanges inside one wavefront need only a
// compiler fence (one wavefront's DS instructions execute in order).
template <bool BLOCK>
__device__ __forceinline__ void exchange_sync()
{
    if constexpr (BLOCK) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

template <class G, int J, bool FIRST_ONLY = false>
__device__ __forceinline__ void load_twiddles(int t, const cf* __restrict__ twN,
                                              cf (&tw)[G::NPASS - 1][G::P - 1])
{
    if constexpr (J < G::NPASS && !(FIRST_ONLY && J > 1)) {
#pragma unroll
        for (int r = 1; r < G::P; ++r) tw[J - 1][r - 1] = twN[twiddle_index<G, J>(t, r)];
        load_twiddles<G, J + 1, FIRST_ONLY>(t, twN, tw);
    }
}

// Twiddles of the passes J >= 2 can live in a small LDS table instead of registers
// (TWLDS): pass J needs W_{L_{J-1}}^{m r} for m < L_J, r < P -- for N = 4096 that
// is 16 x 15 values for pass 2 -- read back with P-1 ds_read_b64 per frame.  Frees
// 2(P-1) VGPRs per such pass.
template <class G, int J>
constexpr int twlds_offset()     // first entry of pass J's block (in cf units)
{
    if constexpr (J <= 2) return 0;
    else return twlds_offset<G, J - 1>() + G::Lcur(J - 1) * (G::P - 1);
}
template <class G>
constexpr int twlds_entries() { return twlds_offset<G, G::NPASS>(); }

template <class G, int J>
__device__ __forceinline__ void fill_twlds(int tid, int nthreads, const cf* __restrict__ twN, cf* table)
{
    if constexpr (J < G::NPASS) {
        if constexpr (J >= 2) {
            constexpr int L = G::Lcur(J), n = L * (G::P - 1);
            for (int i = tid; i < n; i += nthreads) {
                const int m = i / (G::P - 1), r = i % (G::P - 1) + 1;
                table[twlds_offset<G, J>() + i] = twN[m * r * ipow(G::P, J - 1)];
            }
        }
        fill_twlds<G, J + 1>(tid, nthreads, twN, table);
    }
}

// Passes J .. NPASS-1: [fetch] -> radix-P butterfly -> twiddle -> store -> sync.
// Pass J > 1 reads and writes the same LDS slots per thread.  The exchange after
// pass J stays inside groups of L_J threads, so it needs a workgroup barrier
// only if L_J > 64.
// ABL (measurement-only ablations, tuning variants): bit 1 = no butterfly/twiddle
// arithmetic, bit 2 = no LDS stores/fetches (barriers stay).  Results are garbage.
template <class G, int J, int ABL = 0, bool TWLDS = false>
__device__ __forceinline__ void middle_passes(int t, cf* x,
                                              const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab,
                                              PhaseClock& clk, const cf* twtable = nullptr)
{
    if constexpr (J < G::NPASS) {
        if constexpr (J > 1) {
            if constexpr (!(ABL & 4)) phase_fetch<G, J>(t, x, slab);
            asm volatile("" : "+v"(x[0]));   // (timing builds: keep the fetch before the stamp)
            RPF_STAMP(clk, 4 * J);
        }
        if constexpr (!(ABL & 2)) {
            if constexpr (TWLDS && J >= 2) {
                cf twj[G::P - 1];
                const cf* row = twtable + twlds_offset<G, J>() + (t % G::Lcur(J)) * (G::P - 1);
#pragma unroll
                for (int r = 0; r < G::P - 1; ++r) twj[r] = row[r];
                phase_butterfly_twiddle<G>(x, twj);
            } else {
                phase_butterfly_twiddle<G>(x, tw[J - 1]);
            }
        }
        RPF_STAMP(clk, 4 * J + 1);
        if constexpr (!(ABL & 4)) phase_store<G, J>(t, x, slab);
        RPF_STAMP(clk, 4 * J + 2);
        exchange_sync<(G::Lcur(J) > 64)>();
        RPF_STAMP(clk, 4 * J + 3);
        middle_passes<G, J + 1, ABL, TWLDS>(t, x, tw, slab, clk, twtable);
    }
}
template <class G, int J>
__device__ __forceinline__ void middle_passes(int t, cf* x,
                                              const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab)
{
    PhaseClock none;
    middle_passes<G, J>(t, x, tw, slab, none);
}

}  // namespace
}  // namespace rpf
