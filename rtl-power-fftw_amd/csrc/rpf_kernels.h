// rpf_kernels.h -- internal C++ interface between the engine (rpf_engine.cpp)
// and the gfx950 kernels (rpf_kernels.hip).  Not part of the C-ABI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

#include "fft_core.h"
#include "hop_partition.h"

namespace rpf {

struct LaunchInfo {
    int grid = 0;        // workgroups
    int block = 0;       // threads per workgroup
    int fpw = 0;         // frames processed concurrently by one workgroup
    int lds_bytes = 0;   // dynamic LDS per workgroup
    bool partial_f32 = false;   // partial spectra are float32 (tuning variants), else f64
    int slots = 0;       // partial spectra written (0: one per workgroup)
};

// Is there a fused kernel for N bins (tuning variant vid, 0 = default)?
bool kernel_supported(int N, int vid = 0);

// Persistent-grid geometry for N on `device`: li->grid is the number of
// workgroups that are simultaneously resident (occupancy x CU count).
hipError_t plan_launch(int N, int vid, bool window, bool use_dma, int device, LaunchInfo* li);

// Fused unpack + FFT + |X|^2 accumulate over frames [0, nframes) of d_stream
// (frame f = bytes [2N f, 2N (f+1))).  Writes one partial spectrum of N doubles
// per workgroup to d_partial (every workgroup writes, zeros included).
// `grid` is the number of workgroups to launch (<= the planned grid).
hipError_t launch_fft_accum(int N, int vid, bool window, bool use_dma, const uint8_t* d_stream,
                            long nframes, const cf* d_twiddles, const float* d_window,
                            double* d_partial, int grid, hipStream_t stream, LaunchInfo* li);
// The same over the hops of `hops` (hop_partition.h: frame f of hop h = bytes [2N f, 2N (f+1)) of
// hops.stream[h]) in ONE launch.  Workgroup w writes one partial spectrum of N doubles per hop it
// touches, at slot hops.slot_bias[h] + w of d_partial.  `grid` is what partition_hops returned
// for (li->fpw, the planned grid).
hipError_t launch_fft_accum_hops(int N, int vid, bool window, bool use_dma, const HopArgs& hops,
                                 const cf* d_twiddles, const float* d_window, double* d_partial, int grid,
                                 hipStream_t stream, LaunchInfo* li);

// d_out[bin] = (accumulate ? d_out[bin] : 0) + sum_{s < nslots} d_partial[s*stride + bin],
// summed in a fixed order (deterministic).
// slot_stride = distance between partial spectra in elements (0: N).  N even.
// d_skip (may be null): a device word read when the kernel runs; non-zero = leave d_out untouched (the partial
// spectra are not a result: fourstep_fused_abort_word).
hipError_t launch_reduce(const double* d_partial, int nslots, int N, double* d_out,
                         bool accumulate, hipStream_t stream, bool partial_f32 = false, size_t slot_stride = 0,
                         const unsigned* d_skip = nullptr);
// The same for H hops in one launch: d_out[h*N + bin] from the slots [slots.begin[h], slots.begin[h+1]).
hipError_t launch_reduce_hops(const double* d_partial, const SlotRanges& slots, int H, int N, double* d_out,
                              bool accumulate, hipStream_t stream, bool partial_f32 = false, size_t slot_stride = 0,
                              const unsigned* d_skip = nullptr);

// ---- mixed-radix path (KM, rpf_mixed.hip): the planned kernel for the sizes of mixed_plans.inc, its split form
// for those of mixed_plans_split.inc (the two tables are the list of sizes), the run-time Stockham kernel for every
// other even N <= 5120 with prime factors 2, 3, 5 only --
// variant: 0 = the shipped kernel of the size; others (tuning build only) = alternative plans
bool mixed_supported(int N, int variant = 0);
hipError_t plan_mixed(int N, int variant, bool windowed, int device, LaunchInfo* li);
// d_twN: master twiddles W_N^k; one partial spectrum of N doubles per workgroup
hipError_t launch_mixed(int N, int variant, const uint8_t* d_stream, long nframes, const cf* d_twN, const float* d_window,
                        double* d_partial, int grid, hipStream_t stream, LaunchInfo* li);

// ---- Bluestein path (KB in rpf_kernels.hip): any other even N <= 4096 ----------
bool bluestein_supported(int N);
hipError_t plan_bluestein(int N, int device, LaunchInfo* li);
// d_twM: master twiddles of length M = bluestein_length(N); d_g / d_bhat: bluestein_tables.h
hipError_t launch_bluestein(int N, const uint8_t* d_stream, long nframes, const cf* d_twM,
                            const cf* d_g, const cf* d_bhat, double* d_partial, int grid,
                            hipStream_t stream, LaunchInfo* li);

// ---- four-step path (rpf_fourstep.hip): N = N1 x N2, powers of two 16384..262144 --
bool fourstep_supported(int N);
size_t fourstep_scratch_bytes(int N);  // the LARGEST intermediate of one launch pair: 2 GB of complex floats, tile-major (rpf_fourstep.hip)
size_t fourstep_scratch_bytes_per_frame(int N);   // the engine grows its scratch to what its launches need, up to the above
int fourstep_partial_slots(int N);     // partial spectra written by K2b (frame groups)
int fourstep_sub_lengths(int N, int* n1, int* n2);
// Host tables in the kernels' lane order: step_tw[n2][.] = W_N^{n2 k1} (N entries),
// window_t[n2][n1] = window[N2 n1 + n2] (empty without a window).
void fourstep_tables(int N, const float* window, std::vector<cf>& step_tw, std::vector<float>& window_t);
hipError_t fourstep_prepare(int N, int device, LaunchInfo* li);
// Frames [0, nframes) -> d_partial[slots][N] (overwritten); K3 then sums the slots.
// d_tw_n1 / d_tw_n2: master twiddle tables of the two sub-transform lengths;
// d_twN / d_window: fourstep_tables' step_tw / window_t.
hipError_t launch_fourstep(int N, bool window, bool use_dma, const uint8_t* d_stream, long nframes,
                           const cf* d_tw_n1, const cf* d_tw_n2, const cf* d_twN, const float* d_window,
                           cf* d_scratch, size_t scratch_bytes, double* d_partial, int max_grid, hipStream_t stream);

// ---- fused four-step (rpf_fourstep.hip): one persistent kernel, Y stays in each XCD's L2 --------
size_t fourstep_fused_scratch_bytes(int N);   // 32 MB: two 2 MB rounds of Y per XCD
int fourstep_fused_slots(int N);              // partial spectra written: 8 teams x frames per round
size_t fourstep_fused_ctl_bytes();
// fails (hipErrorInvalidValue) unless the device has 256 CUs and one workgroup fits a CU
hipError_t fourstep_fused_prepare(int N, int device, int* grid);
// fault (tests only, rpf_debug_fused_fault): 1 = the launch finds 33 workgroups on XCD 0 (gives up at once);
// 2 = a squatter workgroup holds one CU's LDS until the launch has given up (the real failure: ~0.5 - 5 s)
hipError_t launch_fourstep_fused(int N, bool window, bool use_dma, const uint8_t* d_stream, long nframes,
                                 const cf* d_tw_n1, const cf* d_tw_n2, const cf* d_twN, const float* d_window,
                                 cf* d_scratch, double* d_partial, void* d_ctl, hipStream_t stream, int fault = 0);
// The launch's abort flag as a device word (K3's d_skip): valid from the launch until the next one on the same d_ctl.
const unsigned* fourstep_fused_abort_word(const void* d_ctl);
// After K3: what became of the fused launch.  If it raised its abort flag (teams did not assemble): *verdict = 1 and
// ++*aborts (words the host can read: pinned, mapped), and d_out (may be null: K3 was told to skip) is NaN-filled
// -- what a launch that gave up leaves must not look like a spectrum.
hipError_t launch_fused_verdict(const void* d_ctl, double* d_out, int N, unsigned* verdict, unsigned* aborts,
                                hipStream_t stream);
hipError_t fourstep_fused_aborted(const void* d_ctl, hipStream_t stream, bool* aborted);

// ---- large Bluestein path (rpf_fourstep.hip): even N in (4096, 131072], not a power of two --
bool bigblu_supported(int N);
int bigblu_lengths(int N, int* M, int* m1, int* m2);   // M = m1 * m2 = 2^ceil(log2(2N-1))
size_t bigblu_scratch_bytes(int N);    // two intermediates of 2 GB at most
size_t bigblu_scratch_bytes_per_frame(int N);
int bigblu_partial_slots(int N);       // partial spectra of M (not N) doubles each
hipError_t bigblu_prepare(int N, int device, LaunchInfo* li);
// Host tables (each M entries, in the kernels' lane order) from bluestein_tables.h's g (N) and bhat (M).
void bigblu_tables(int N, const cf* g, const cf* bhat, std::vector<cf>& g_t, std::vector<cf>& bhat_t,
                   std::vector<cf>& step_tw, std::vector<cf>& step_tw2);
// d_tw_m1 / d_tw_m2: master twiddles of lengths m1, m2; the rest: bigblu_tables' outputs.
hipError_t launch_bigblu(int N, bool use_dma, const uint8_t* d_stream, long nframes, const cf* d_tw_m1,
                         const cf* d_tw_m2, const cf* d_step_tw, const cf* d_step_tw2, const cf* d_g_t,
                         const cf* d_bhat_t, cf* d_scratch, size_t scratch_bytes, double* d_partial, int max_grid,
                         hipStream_t stream);

// ---- generic path (rpf_generic.hip): every other even N -- powers of two up to 2^26, others up to 2^23 --
bool generic_supported(int N);
int generic_length(int N);              // N, or Bluestein's M = 2^ceil(log2(2N-1))
int generic_batch(int N);
size_t generic_scratch_bytes(int N);
void generic_twiddle_tables(int N, std::vector<cf>& t0, std::vector<cf>& t1, int* h);
// d_pwr[N] = (accumulate ? d_pwr : 0) + sum over the frames; d_window: N floats or null (powers of two);
// d_g (N) / d_bhat (M): bluestein_tables.h's tables (other N).
hipError_t launch_generic(int N, const uint8_t* d_stream, long nframes, const float* d_window, const cf* d_g,
                          const cf* d_bhat, const cf* d_t0, const cf* d_t1, int h, cf* d_scratch, double* d_pwr,
                          bool accumulate, hipStream_t stream);

// Master twiddle table W_N^k = exp(-2 pi i k / N), k in [0,N), evaluated in
// long double and rounded once to float.
void make_twiddles(int N, std::vector<cf>& out);

}  // namespace rpf
