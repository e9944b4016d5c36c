// rpf_device_common.h -- device-side helpers shared by the gfx950 kernels
// (rpf_kernels.hip: LDS-resident FFT sizes; rpf_fourstep.hip: four-step sizes).
#pragma once

#include <hip/hip_runtime.h>

#include "fft_core.h"

namespace rpf {
namespace {

// Orders LDS traffic between the threads that exchange data: a workgroup
// barrier when a frame spans several wavefronts, otherwise only a compiler
// fence (one wavefront's DS instructions execute in order).
template <bool BLOCK>
__device__ __forceinline__ void exchange_sync()
{
    if constexpr (BLOCK) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

template <class G, int J>
__device__ __forceinline__ void load_twiddles(int t, const cf* __restrict__ twN,
                                              cf (&tw)[G::NPASS - 1][G::P - 1])
{
    if constexpr (J < G::NPASS) {
#pragma unroll
        for (int r = 1; r < G::P; ++r) tw[J - 1][r - 1] = twN[twiddle_index<G, J>(t, r)];
        load_twiddles<G, J + 1>(t, twN, tw);
    }
}

// Passes J .. NPASS-1: [fetch] -> radix-P butterfly -> twiddle -> store -> sync.
// Pass J > 1 reads and writes the same LDS slots per thread.  The exchange after
// pass J stays inside groups of L_J threads, so it needs a workgroup barrier
// only if L_J > 64.
template <class G, int J>
__device__ __forceinline__ void middle_passes(int t, cf* x,
                                              const cf (&tw)[G::NPASS - 1][G::P - 1], cf* slab)
{
    if constexpr (J < G::NPASS) {
        if constexpr (J > 1) phase_fetch<G, J>(t, x, slab);
        phase_butterfly_twiddle<G>(x, tw[J - 1]);
        phase_store<G, J>(t, x, slab);
        exchange_sync<(G::Lcur(J) > 64)>();
        middle_passes<G, J + 1>(t, x, tw, slab);
    }
}

}  // namespace
}  // namespace rpf
