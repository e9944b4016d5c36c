/*
 * rpf_engine.h -- C-ABI of the MI355X power-spectrum engine.
 *
 * This is the drop-in boundary for the FFT-and-accumulate worker of
 * rtl_power_fftw: everything `class Datastore` exposes to its two callers
 * (/root/reference/src/datastore.h:35-68, used from
 * /root/reference/src/acquisition.cxx:252-256,278-324,343-347,377-397 and
 * /root/reference/src/rtl_power_fftw.cxx:112,169,215), flattened to
 * `extern "C"` functions on an opaque handle, plain pointers and sizes.
 * No exceptions cross it; every call returns an int that is either RPF_OK or
 * one of the reference's own process exit codes
 * (/root/reference/src/exceptions.h:25-34), so the host turns a failure into
 * `throw RPFexception(rpf_last_error(e), (ReturnValue)rc)` unchanged.
 *
 * Threading contract (same as the reference, datastore.h:40-47): exactly one
 * producer thread calls begin/acquire/submit/unget/finish; the engine owns its
 * consumer thread and its HIP streams; rpf_get_* are valid after rpf_finish.
 *
 * The implementation (rtl-power-fftw_amd/csrc) is hand-written HIP for gfx950.
 * There is no CPU fallback: without a usable HIP device rpf_engine_create
 * fails with RPF_ERR_HARDWARE.
 */
#ifndef RPF_ENGINE_H
#define RPF_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPF_ABI_VERSION 2   /* 2: rpf_accumulate_device_hops, rpf_device_fused_hops, rpf_scan_reducer_* */

/* Return codes = ReturnValue of /root/reference/src/exceptions.h:25-34. */
#define RPF_OK 0
#define RPF_ERR_INVALID_ARGUMENT 3 /* ReturnValue::InvalidArgument */
#define RPF_ERR_INVALID_INPUT 5    /* ReturnValue::InvalidInput    */
#define RPF_ERR_ACQUISITION 6      /* ReturnValue::AcquisitionError */
#define RPF_ERR_HARDWARE 7         /* ReturnValue::HardwareError   */

typedef struct rpf_engine rpf_engine;

/* Mirrors the Params fields Datastore reads (datastore.cxx:23-34,67-76):
 * N, buffers, buf_length, window (+ the window values of AuxData). */
typedef struct rpf_config {
    uint32_t struct_size;     /* = sizeof(rpf_config), for ABI evolution          */
    int32_t N;                /* params.N: FFT bins, even (params.cxx:150-155)    */
    const float* window;      /* params.window ? N floats (copied) : NULL         */
    int32_t n_buffers;        /* params.buffers (default 5, params.h:42)          */
    int64_t buffer_capacity;  /* params.buf_length in bytes (params.h:43)         */
    int32_t device;           /* HIP device ordinal                               */
    uint32_t flags;           /* RPF_FLAG_*                                       */
} rpf_config;

#define RPF_FLAG_NONE 0u
/* Stage raw bytes through VGPRs instead of LDS-DMA (debug / A-B measurement). */
#define RPF_FLAG_NO_LDS_DMA 1u
/* Sizes 16384..262144 (the four-step sizes): the fused persistent kernel -- ONE launch per acquisition, the
 * intermediate is handed from the column transforms to the row transforms inside each XCD's L2 instead of crossing
 * the fabric between two kernels.  It is what these sizes run by default on a 256-CU part (65536 ... 262144; 16384 and
 * 32768 default to the LDS mixed-radix kernels -- windowed runs of 32768 excepted, which are faster here -- so asking for
 * it is also asking for the four-step path) whenever
 * its eight workgroup teams assemble at rpf_engine_create; otherwise the engine keeps the two-kernel path.  A launch
 * whose teams do not assemble later (a CU held by someone else's kernel for seconds: another process on the device)
 * gives up, and the engine leaves the fused kernel for the rest of its life:
 *   - buffer-queue path (rpf_begin .. rpf_finish, rpf_accumulate): the acquisition does NOT fail -- like the
 *     reference's worker (datastore.cxx:48-96) this one has no failure path for it.  K3 leaves the accumulator alone,
 *     the worker runs the same staged bytes through the two-kernel path and rpf_finish returns RPF_OK with the spectrum
 *     that path gives (bit-identical to RPF_FLAG_NO_FOURSTEP_FUSED whenever every fused launch in flight gave up,
 *     else equal up to the order of the double additions);
 *   - device-resident entries (rpf_accumulate_device*, rpf_device_*; they return without synchronising): d_pwr_out of
 *     THAT launch is NaN-filled -- loud, never wrong -- and the next entry (or rpf_fused_status) that runs after the
 *     caller's stream has passed the launch notices and switches to the two-kernel path; see rpf_fused_status.
 *  */
#define RPF_FLAG_FOURSTEP_FUSED 2u
/* The four-step sizes on the two-kernel path (intermediate through HBM) even where the fused kernel is available
 * (A/B measurement, and the fallback's own tests). */
#define RPF_FLAG_NO_FOURSTEP_FUSED 8u
/* Sizes served by the LDS mixed-radix kernels (the tables csrc/mixed_plans.inc and
 * csrc/mixed_plans_split.inc: 500, 1000, 3000 ... 80000; 16384, 32768): use the kernel they would
 * get without them -- Bluestein, resp. the four-step pair for 16384 and 32768 (A/B measurement;
 * all are exact to the float32 bar). */
#define RPF_FLAG_NO_MIXED_RADIX 4u
/* Tuning: select kernel variant k for this N.  The shipped library contains only
 * variant 0 (one kernel per N x {window} x {staging}); any other k makes
 * rpf_engine_create fail with RPF_ERR_INVALID_ARGUMENT.  Experimental variants
 * exist only in the separate -DRPF_TUNING build used by tools/. */
#define RPF_FLAG_VARIANT(k) (((uint32_t)(k) & 0xffu) << 8)

/* ABI version of the loaded library. */
int rpf_abi_version(void);
/* 1 if this build has a gfx950 kernel for N bins, else 0 (the supported set is
 * listed in DESIGN.md; an unsupported N makes rpf_engine_create fail with
 * RPF_ERR_INVALID_ARGUMENT rather than fall back to anything). */
int rpf_supported_n(int N);
/* Message of the last failure on this thread when no engine exists yet. */
const char* rpf_last_global_error(void);

/* Datastore::Datastore (datastore.cxx:23-34): buffer pool (pinned host memory
 * the producer fills directly), FFT plan (= twiddle tables on the device),
 * zeroed pwr[N] and queue_histogram[n_buffers+1]. */
int rpf_engine_create(const rpf_config* cfg, rpf_engine** out);
/* Datastore::~Datastore (datastore.cxx:36-46). */
void rpf_engine_destroy(rpf_engine* e);
const char* rpf_last_error(const rpf_engine* e);

/* Start of Acquisition::run's worker section (acquisition.cxx:252-256):
 * pwr := 0, acquisition_finished := false, repeats_done := 0, start the
 * consumer.  `repeats` = params.repeats for this acquisition. */
int rpf_begin(rpf_engine* e, int64_t repeats);

/* acquisition.cxx:278-285: samples queue_histogram[#empty] and then blocks
 * until a buffer is free; returns it with its capacity. */
int rpf_buffer_acquire(rpf_engine* e, uint8_t** buf, size_t* capacity);
/* acquisition.cxx:302,320-323: buffer.resize(nbytes) + push_back to
 * occupied_buffers + notify.  nbytes even, <= capacity. */
int rpf_buffer_submit(rpf_engine* e, uint8_t* buf, size_t nbytes);
/* acquisition.cxx:310-314: failed readout, buffer goes back to the FRONT of
 * empty_buffers. */
int rpf_buffer_unget(rpf_engine* e, uint8_t* buf);

/* acquisition.cxx:343-347: acquisition_finished := true, notify, join.  On
 * return every submitted byte has been consumed per datastore.cxx:67-89
 * (frames may straddle buffers; frames beyond `repeats` and a trailing partial
 * frame are dropped) and pwr/repeats_done are final. */
int rpf_finish(rpf_engine* e, int64_t* repeats_done);

/* Datastore::pwr (datastore.h:53) -- raw accumulated |X|^2 per bin, bin N/2 =
 * DC; the DC interpolation of acquisition.cxx:377 is the caller's.  */
int rpf_get_power(const rpf_engine* e, double* out /* N */);
/* Datastore::repeats_done (datastore.h:38). */
int64_t rpf_get_repeats_done(const rpf_engine* e);
/* Datastore::queue_histogram (datastore.h:47), cumulative over the engine's
 * life like the reference's (never reset, datastore.cxx:24). */
int rpf_get_histogram(const rpf_engine* e, int* out /* n_buffers + 1 */);

/* Whole acquisition on one contiguous host stream, driven through the same
 * begin/acquire/submit/finish path in buffer_capacity-sized pieces. */
int rpf_accumulate(rpf_engine* e, const uint8_t* stream, size_t nbytes, int64_t repeats,
                   double* pwr_out /* N, host */, int64_t* repeats_done);

/* Replay without the copy into the pool: pin a caller-owned stream (hipHostRegister) so that rpf_accumulate on any part of
 * it hands the bytes to the consumer where they lie -- one H2D copy per staging slot instead of memcpy + copy per buffer
 * (measured: 9 -> 25 Gsample/s with the reference's default buffers).  Meant for a stream that is replayed more than once:
 * pinning costs ~4 ms per 82 MB and the FIRST copy out of freshly pinned memory runs at ~3 GB/s (the IOMMU mappings are
 * made then); rpf_accumulate never pins by itself.  The caller keeps the memory alive until rpf_stream_unregister or
 * rpf_engine_destroy (which unpins what is still registered).  RPF_ERR_HARDWARE if the runtime cannot pin the range (already
 * pinned memory, for one). */
int rpf_stream_register(rpf_engine* e, const void* stream, size_t nbytes);
int rpf_stream_unregister(rpf_engine* e, const void* stream);

/* Device-resident replay: the stream already sits in HBM (d_stream: even
 * address required, else RPF_ERR_INVALID_ARGUMENT; 16-byte aligned for the
 * LDS-DMA staging path, other alignments silently stage through VGPRs).  The
 * engine's device is made current for the call and the caller's restored.  Enqueues the fused kernel and the partial-sum reduce on
 * `hip_stream` (a hipStream_t; NULL = HIP's null stream) and returns
 * without synchronising (one exception: the two-kernel four-step and large Bluestein paths grow their intermediate to
 * what a launch needs -- a call that needs more than any before it synchronises `hip_stream` once, frees and
 * allocates); d_pwr_out[N] (device doubles, 16-byte aligned -- the reduce stores
 * bin pairs -- else RPF_ERR_INVALID_ARGUMENT; the same holds for every d_pwr_out below) is
 * overwritten with the sum over frames [0, min(repeats, nbytes/(2N))).  Used by bench.py and the
 * full-size parity tests; does not touch the buffer queues. */
int rpf_accumulate_device(rpf_engine* e, const void* d_stream, size_t nbytes, int64_t repeats,
                          double* d_pwr_out, void* hip_stream, int64_t* repeats_done);

/* The two halves of rpf_accumulate_device as separate enqueues, so that a
 * benchmark can bracket the dominant kernel alone with events on `hip_stream`:
 * _fused runs K1 (unpack+FFT+|X|^2) and leaves one partial spectrum per frame
 * slot in engine scratch; _reduce runs K3 over the scratch of the last _fused
 * call into d_pwr_out[N] (after rpf_device_fused_hops: all n_hops spectra, d_pwr_out[n_hops x N]). */
int rpf_device_fused(rpf_engine* e, const void* d_stream, size_t nbytes, int64_t repeats,
                     void* hip_stream, int64_t* repeats_done);
int rpf_device_reduce(rpf_engine* e, double* d_pwr_out, void* hip_stream);

/* A whole scan in one call: n_hops device-resident acquisitions (the reference's scan is a loop of
 * hops, /root/reference/src/rtl_power_fftw.cxx:133-174, each starting from a zeroed accumulator,
 * acquisition.cxx:252-254).  Hop h = the first min(repeats[h], nbytes[h]/(2N)) frames of
 * d_streams[h]; its spectrum goes to d_pwr_out[h*N .. h*N+N) (device doubles, overwritten; zeros
 * for a hop without a whole frame).  For the sizes the LDS-resident kernel serves (powers of two
 * 64 .. 8192) up to rpf_max_hops_per_launch() hops share ONE persistent kernel launch -- the
 * workgroups walk the hops' frames as one sequence and hand over / zero their register
 * accumulators at each hop boundary -- and ONE reduce launch, so a scan pays the per-launch
 * fixed cost once instead of once per hop; other sizes run hop by hop.  Same stream, alignment
 * and synchronisation rules as rpf_accumulate_device.  repeats_done: n_hops entries or NULL. */
int rpf_accumulate_device_hops(rpf_engine* e, const void* const* d_streams, const size_t* nbytes,
                               const int64_t* repeats, int n_hops, double* d_pwr_out /* n_hops x N */,
                               void* hip_stream, int64_t* repeats_done);
/* The fused-kernel half of it alone (n_hops <= rpf_max_hops_per_launch(), LDS-resident sizes
 * only, else RPF_ERR_INVALID_ARGUMENT); rpf_device_reduce then writes all n_hops spectra. */
int rpf_device_fused_hops(rpf_engine* e, const void* const* d_streams, const size_t* nbytes,
                          const int64_t* repeats, int n_hops, void* hip_stream, int64_t* repeats_done);
int rpf_max_hops_per_launch(void);

/* Datastore::pwr as it sits in HBM after rpf_finish: copied (device to device, or peer to peer when
 * dst_device is another device) into d_dst[N]; synchronises hip_stream before returning. */
int rpf_copy_power_device(const rpf_engine* e, double* d_dst, void* hip_stream, int dst_device);

/* ---- multi-GPU scans in one process: the final reduce over RCCL / xGMI ----------------
 * One engine per device runs its share of a scan's hops (hop-major, frame-aligned); a scan reducer owns
 * one RCCL communicator over those devices (ncclCommInitAll; librccl.so is loaded with dlopen) and, on
 * every device, a block of max_hops x N doubles.  Per pass: _begin zeroes the blocks; after an engine's
 * rpf_finish, _deposit copies its accumulator into row `hop` of its device's block; _reduce issues ONE
 * ncclReduce(sum, ncclDouble, hops x N) onto the first device and one device-to-host copy.
 * rpf_scan_reducer_create fails with RPF_ERR_HARDWARE when RCCL is missing or refuses the device list
 * (e.g. one device listed twice); callers then add the per-device spectra on the host, which gives the
 * same sums up to the order of the additions. */
typedef struct rpf_scan_reducer rpf_scan_reducer;
int rpf_scan_reducer_create(const int* devices, int n_devices, int N, int max_hops, rpf_scan_reducer** out);
void rpf_scan_reducer_destroy(rpf_scan_reducer* r);
const char* rpf_scan_reducer_last_error(const rpf_scan_reducer* r);
int rpf_scan_reducer_begin(rpf_scan_reducer* r);
int rpf_scan_reducer_deposit(rpf_scan_reducer* r, int slot, int hop, const rpf_engine* e);
int rpf_scan_reducer_reduce(rpf_scan_reducer* r, int hops, double* host_out /* hops x N */);

/* Which four-step kernel this engine runs, and what became of its fused launches.
 *   *active            1: the fused persistent kernel is what the next launch runs; 0: the two-kernel path (or N is not
 *                      a four-step size)
 *   *launches_gave_up  fused launches whose teams did not assemble, over the engine's life.  Device-resident entries:
 *                      read it after synchronising the stream; a count that has grown means the spectrum of that
 *                      launch is NaN and must be asked for again (the engine is on the two-kernel path by then).
 *   *launches_recovered  of those, the ones the buffer-queue worker ran again on the two-kernel path
 * Any pointer may be NULL.  Not to be called while an acquisition is running.  NOT a pure query (hence no const): a call
 * that finds a device-resident launch to have given up retires the fused kernel for this engine then and there. */
int rpf_fused_status(rpf_engine* e, int* active, int64_t* launches_gave_up, int64_t* launches_recovered);

/* Launch geometry of the last fused-kernel launch (for DESIGN/bench reporting):
 * workgroups, threads per workgroup, frames per workgroup, LDS bytes. */
int rpf_last_launch_info(const rpf_engine* e, int* grid, int* block, int* frames_per_wg,
                         int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RPF_ENGINE_H */
