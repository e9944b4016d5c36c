/*
 * rpf_oracle.h -- CPU restatement of the rtl_power_fftw FFT-and-accumulate worker.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
 * load or call it, and there only as the checker / reported CPU baseline.  The
 * product path (rtl-power-fftw_amd/csrc, behind include/rpf_engine.h) never
 * links or calls this code and has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures
 * for this path, and its arithmetic core is FFTW3 single precision (fftw3f,
 * version unpinned, /root/reference/CMakeLists.txt:7, planned with
 * FFTW_MEASURE at /root/reference/src/datastore.cxx:32-33), which is absent
 * from this image; the reference's worker therefore cannot be built here
 * (no fftw3.h, no TCLAP, no librtlsdr) and there is no oracle/_ref.  This file
 * restates the reference loops exactly and replaces fftwf_execute by its
 * published definition (forward, unnormalised, single-precision DFT computed
 * by a Cooley-Tukey/Stockham FFT).  It is pinned instead against float64
 * truth (numpy complex128, tests/golden/), analytic known answers and the
 * output-format example of /root/reference/doc/rtl_power_fftw.1.md:94-99.
 *
 * Every function cites the reference lines it follows.
 */
#ifndef RPF_ORACLE_H
#define RPF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- FFT provider: stands where fftwf_plan_dft_1d/fftwf_execute stand ----
 * /root/reference/src/datastore.cxx:30-33 (plan: N-point, forward, c2c,
 * out-of-place, unnormalised) and :82 (execute).  Any N >= 1. */
typedef struct rpf_oracle_plan rpf_oracle_plan;
rpf_oracle_plan* rpf_oracle_plan_create(int N);
void rpf_oracle_plan_destroy(rpf_oracle_plan* p);
/* in/out: N interleaved (re,im) pairs; in and out must not alias. */
void rpf_oracle_fft_f32(const rpf_oracle_plan* p, const float* in, float* out);
void rpf_oracle_fft_f64(const rpf_oracle_plan* p, const double* in, double* out);

/* ---- The worker: Datastore::fftThread minus the mutex/condvar plumbing ----
 * /root/reference/src/datastore.cxx:48-96.  precision: 32 = the reference's
 * arithmetic (float FFT, double accumulate); 64 = float64 truth variant
 * (unpack and window rounding identical to the reference, FFT in double). */
typedef struct rpf_oracle_worker rpf_oracle_worker;
rpf_oracle_worker* rpf_oracle_worker_create(int N, const float* window /* N floats or NULL */,
                                            int precision);
void rpf_oracle_worker_destroy(rpf_oracle_worker* w);
/* /root/reference/src/acquisition.cxx:252-254: pwr := 0, repeats_done := 0;
 * and the worker-local fft_pointer := 0 (datastore.cxx:52). */
void rpf_oracle_worker_begin(rpf_oracle_worker* w, int64_t repeats);
/* One occupied buffer through the loop at datastore.cxx:65-89. */
void rpf_oracle_worker_consume(rpf_oracle_worker* w, const uint8_t* buf, size_t nbytes);
int64_t rpf_oracle_worker_repeats_done(const rpf_oracle_worker* w);
const double* rpf_oracle_worker_pwr(const rpf_oracle_worker* w);

/* Whole acquisition over one contiguous stream (convenience for tests/bench). */
int rpf_oracle_accumulate(int N, const float* window, int precision,
                          const uint8_t* stream, size_t nbytes, int64_t repeats,
                          double* pwr_out /* N */, int64_t* repeats_done_out);

/* Same, frames [0,repeats) split over nthreads disjoint contiguous ranges and
 * summed in thread order.  Not the reference's structure (it has exactly one
 * FFT thread, acquisition.cxx:256); used only as the all-cores CPU baseline. */
int rpf_oracle_accumulate_mt(int N, const float* window, const uint8_t* stream, size_t nbytes,
                             int64_t repeats, int nthreads, double* pwr_out,
                             int64_t* repeats_done_out);
/* The same, timing-friendly: one persistent worker (plan) per thread, `loops` walks over its frame range;
 * *seconds_in_threads = wall time from the first thread's start to the last one's end (planning excluded). */
int rpf_oracle_accumulate_mt_loops(int N, const float* window, const uint8_t* stream, size_t nbytes,
                                   int64_t repeats, int nthreads, int loops, double* pwr_out,
                                   int64_t* repeats_done_out, double* seconds_in_threads);

/* ---- Output stage: Acquisition::write_data, text mode ----
 * /root/reference/src/acquisition.cxx:360-433.  Mutates pwr[N/2] (DC
 * interpolation, :377) exactly as the reference does.  Writes the data lines
 * only (from the first "freq value" line through the trailing blank line);
 * the five '#' header lines carry wall-clock timestamps and are the caller's.
 * Returns bytes written (excluding NUL) or -1 if cap is too small. */
long rpf_oracle_format_text(double* pwr, int N, int64_t repeats_done, int64_t tuned_freq,
                            int samplerate, int linear, const double* baseline /* or NULL */,
                            char* out, size_t cap);
/* Matrix mode row (acquisition.cxx:400-405): N float32 values. */
void rpf_oracle_format_matrix(double* pwr, int N, int64_t repeats_done, int samplerate,
                              int linear, const double* baseline, float* row_out);

/* ---- Plan: repeats / buffer length / hop list ----
 * /root/reference/src/acquisition.cxx:158-198. */
typedef struct {
    int N;
    int sample_rate;                 /* actual_samplerate */
    int64_t repeats;                 /* in: -n value or default; out: possibly from -t */
    int integration_time_isSet;
    double integration_time;
    int buf_length;                  /* in/out */
    int buf_length_isSet;
    int freq_hopping_isSet;
    int64_t startfreq, stopfreq, cfreq;
    double min_overlap;
} rpf_oracle_plan_params;
/* Returns number of hops written to freqs (<= cap), or -1 if cap too small. */
int rpf_oracle_make_plan(rpf_oracle_plan_params* p, int64_t* freqs, int cap);

/* Producer-side read size (acquisition.cxx:288-300). */
int64_t rpf_oracle_data_needed(int64_t dataTotal, int64_t dataRead, int buf_length);

#ifdef __cplusplus
}
#endif
#endif /* RPF_ORACLE_H */
