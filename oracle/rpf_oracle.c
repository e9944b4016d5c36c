/*
 * rpf_oracle.c -- CPU restatement of the rtl_power_fftw FFT-and-accumulate
 * worker and its immediate neighbours (Plan, write_data).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see rpf_oracle.h for both
 * statements.  Plain C11, no dependencies beyond libc/libm/pthread.
 */
#define _GNU_SOURCE
#include "rpf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ FFT -- */

#define REAL float
#define WIDE double
#define SUFFIX f32
#include "rpf_oracle_fft.inc"
#undef REAL
#undef WIDE
#undef SUFFIX

#define REAL double
#define WIDE long double
#define SUFFIX f64
#include "rpf_oracle_fft.inc"
#undef REAL
#undef WIDE
#undef SUFFIX

struct rpf_oracle_plan {
    int N;
    int nf;
    int factors[64];
    int max_radix;
    cpx_f32* tw32;
    cpx_f64* tw64;
    /* scratch (a plan is used by one thread at a time, like an fftwf_plan
     * bound to its inbuf/outbuf, datastore.cxx:30-33) */
    cpx_f32 *a32, *b32, *t32;
    cpx_f64 *a64, *b64, *t64;
};

static int factorize(int N, int* factors, int* max_radix)
{
    int nf = 0, n = N;
    *max_radix = 1;
    while (n % 4 == 0) { factors[nf++] = 4; n /= 4; }
    while (n % 2 == 0) { factors[nf++] = 2; n /= 2; }
    for (int p = 3; (long long)p * p <= n; p += 2)
        while (n % p == 0) { factors[nf++] = p; n /= p; }
    if (n > 1) factors[nf++] = n;
    for (int i = 0; i < nf; ++i)
        if (factors[i] > *max_radix) *max_radix = factors[i];
    return nf;
}

rpf_oracle_plan* rpf_oracle_plan_create(int N)
{
    if (N < 1) return NULL;
    rpf_oracle_plan* p = (rpf_oracle_plan*)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->N = N;
    p->nf = factorize(N, p->factors, &p->max_radix);
    p->tw32 = make_twiddles_f32(N);
    p->tw64 = make_twiddles_f64(N);
    p->a32 = (cpx_f32*)malloc(sizeof(cpx_f32) * (size_t)N);
    p->b32 = (cpx_f32*)malloc(sizeof(cpx_f32) * (size_t)N);
    p->t32 = (cpx_f32*)malloc(sizeof(cpx_f32) * (size_t)p->max_radix);
    p->a64 = (cpx_f64*)malloc(sizeof(cpx_f64) * (size_t)N);
    p->b64 = (cpx_f64*)malloc(sizeof(cpx_f64) * (size_t)N);
    p->t64 = (cpx_f64*)malloc(sizeof(cpx_f64) * (size_t)p->max_radix);
    if (!p->tw32 || !p->tw64 || !p->a32 || !p->b32 || !p->t32 || !p->a64 || !p->b64 || !p->t64) {
        rpf_oracle_plan_destroy(p);
        return NULL;
    }
    return p;
}

void rpf_oracle_plan_destroy(rpf_oracle_plan* p)
{
    if (!p) return;
    free(p->tw32); free(p->tw64);
    free(p->a32); free(p->b32); free(p->t32);
    free(p->a64); free(p->b64); free(p->t64);
    free(p);
}

void rpf_oracle_fft_f32(const rpf_oracle_plan* p, const float* in, float* out)
{
    fft_run_f32(p->N, p->factors, p->nf, p->tw32, p->a32, p->b32, p->t32, in, out);
}

void rpf_oracle_fft_f64(const rpf_oracle_plan* p, const double* in, double* out)
{
    fft_run_f64(p->N, p->factors, p->nf, p->tw64, p->a64, p->b64, p->t64, in, out);
}

/* --------------------------------------------------------------- worker -- */

struct rpf_oracle_worker {
    int N;
    int precision;
    int has_window;            /* params.window */
    float* window_values;      /* Datastore::window_values, datastore.h:49 */
    int64_t repeats;           /* params.repeats */
    int64_t repeats_done;      /* Datastore::repeats_done, datastore.h:38 */
    int fft_pointer;           /* local of fftThread, datastore.cxx:52 */
    float* inbuf;              /* datastore.h:51 (complex<float>[N]) */
    float* outbuf;
    double* inbuf64;           /* truth variant only */
    double* outbuf64;
    double* pwr;               /* datastore.h:53 */
    rpf_oracle_plan* plan;     /* datastore.h:52 */
};

rpf_oracle_worker* rpf_oracle_worker_create(int N, const float* window, int precision)
{
    /* N even is enforced upstream (params.cxx:150-155); odd N would break the
     * (-1)^n centring trick (datastore.cxx:69-72). */
    if (N < 2 || (N % 2) != 0 || (precision != 32 && precision != 64)) return NULL;
    rpf_oracle_worker* w = (rpf_oracle_worker*)calloc(1, sizeof(*w));
    if (!w) return NULL;
    w->N = N;
    w->precision = precision;
    w->plan = rpf_oracle_plan_create(N);
    w->inbuf = (float*)malloc(sizeof(float) * 2 * (size_t)N);
    w->outbuf = (float*)malloc(sizeof(float) * 2 * (size_t)N);
    w->inbuf64 = (double*)malloc(sizeof(double) * 2 * (size_t)N);
    w->outbuf64 = (double*)malloc(sizeof(double) * 2 * (size_t)N);
    w->pwr = (double*)calloc((size_t)N, sizeof(double));     /* datastore.cxx:25 */
    if (window) {
        w->has_window = 1;
        w->window_values = (float*)malloc(sizeof(float) * (size_t)N);
        if (w->window_values) memcpy(w->window_values, window, sizeof(float) * (size_t)N);
    }
    if (!w->plan || !w->inbuf || !w->outbuf || !w->inbuf64 || !w->outbuf64 || !w->pwr ||
        (window && !w->window_values)) {
        rpf_oracle_worker_destroy(w);
        return NULL;
    }
    return w;
}

void rpf_oracle_worker_destroy(rpf_oracle_worker* w)
{
    if (!w) return;
    rpf_oracle_plan_destroy(w->plan);
    free(w->inbuf); free(w->outbuf); free(w->inbuf64); free(w->outbuf64);
    free(w->pwr); free(w->window_values);
    free(w);
}

void rpf_oracle_worker_begin(rpf_oracle_worker* w, int64_t repeats)
{
    /* acquisition.cxx:252 std::fill(pwr, 0); :254 repeats_done = 0;
     * :256 a fresh fftThread starts with fft_pointer = 0 (datastore.cxx:52). */
    memset(w->pwr, 0, sizeof(double) * (size_t)w->N);
    w->repeats = repeats;
    w->repeats_done = 0;
    w->fft_pointer = 0;
}

void rpf_oracle_worker_consume(rpf_oracle_worker* w, const uint8_t* buffer, size_t size)
{
    const int N = w->N;
    size_t buffer_pointer = 0;                                   /* datastore.cxx:66 */
    while (buffer_pointer < size && w->repeats_done < w->repeats) {     /* :67 */
        while (w->fft_pointer < N && buffer_pointer < size) {           /* :68 */
            /* :73 odd samples are rotated by pi => spectrum shifted by N/2 */
            const float multiplier = (w->fft_pointer % 2 == 0 ? 1.0f : -1.0f);
            /* :74 complex<float>(uint8, uint8)  :75 minus (127.0f,127.0f), times +-1 */
            float re = ((float)buffer[buffer_pointer] - 127.0f) * multiplier;
            float im = ((float)buffer[buffer_pointer + 1] - 127.0f) * multiplier;
            if (w->has_window) {                                        /* :76-77 */
                re *= w->window_values[w->fft_pointer];
                im *= w->window_values[w->fft_pointer];
            }
            w->inbuf[2 * w->fft_pointer] = re;
            w->inbuf[2 * w->fft_pointer + 1] = im;
            buffer_pointer += 2;                                        /* :78 */
            w->fft_pointer++;                                           /* :79 */
        }
        if (w->fft_pointer == N) {                                      /* :81 */
            if (w->precision == 32) {
                rpf_oracle_fft_f32(w->plan, w->inbuf, w->outbuf);       /* :82 */
                for (int i = 0; i < N; ++i) {                           /* :83-85 */
                    /* pow(float,2) promotes to double: both squares are exact,
                     * their sum rounds once, the running sum is double. */
                    const double re = (double)w->outbuf[2 * i];
                    const double im = (double)w->outbuf[2 * i + 1];
                    w->pwr[i] += re * re + im * im;
                }
            } else {
                for (int i = 0; i < 2 * N; ++i) w->inbuf64[i] = (double)w->inbuf[i];
                rpf_oracle_fft_f64(w->plan, w->inbuf64, w->outbuf64);
                for (int i = 0; i < N; ++i) {
                    const double re = w->outbuf64[2 * i];
                    const double im = w->outbuf64[2 * i + 1];
                    w->pwr[i] += re * re + im * im;
                }
            }
            w->repeats_done++;                                          /* :86 */
            w->fft_pointer = 0;                                         /* :87 */
        }
    }
}

int64_t rpf_oracle_worker_repeats_done(const rpf_oracle_worker* w) { return w->repeats_done; }
const double* rpf_oracle_worker_pwr(const rpf_oracle_worker* w) { return w->pwr; }

int rpf_oracle_accumulate(int N, const float* window, int precision, const uint8_t* stream,
                          size_t nbytes, int64_t repeats, double* pwr_out,
                          int64_t* repeats_done_out)
{
    rpf_oracle_worker* w = rpf_oracle_worker_create(N, window, precision);
    if (!w) return -1;
    rpf_oracle_worker_begin(w, repeats);
    rpf_oracle_worker_consume(w, stream, nbytes);
    memcpy(pwr_out, w->pwr, sizeof(double) * (size_t)N);
    if (repeats_done_out) *repeats_done_out = w->repeats_done;
    rpf_oracle_worker_destroy(w);
    return 0;
}

typedef struct {
    int N;
    const float* window;
    const uint8_t* stream;
    int64_t first, count;
    double* pwr;
    int rc;
} mt_job;

static void* mt_run(void* arg)
{
    mt_job* j = (mt_job*)arg;
    int64_t done = 0;
    j->rc = rpf_oracle_accumulate(j->N, j->window, 32, j->stream + 2 * (size_t)j->N * (size_t)j->first,
                                  2 * (size_t)j->N * (size_t)j->count, j->count, j->pwr, &done);
    if (j->rc == 0 && done != j->count) j->rc = -2;
    return NULL;
}

int rpf_oracle_accumulate_mt(int N, const float* window, const uint8_t* stream, size_t nbytes,
                             int64_t repeats, int nthreads, double* pwr_out,
                             int64_t* repeats_done_out)
{
    if (N < 2 || nthreads < 1) return -1;
    int64_t frames = (int64_t)(nbytes / (2 * (size_t)N));
    if (frames > repeats) frames = repeats;
    if (nthreads > frames) nthreads = frames > 0 ? (int)frames : 1;
    mt_job* jobs = (mt_job*)calloc((size_t)nthreads, sizeof(mt_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    double* partial = (double*)calloc((size_t)nthreads * (size_t)N, sizeof(double));
    if (!jobs || !th || !partial) { free(jobs); free(th); free(partial); return -1; }
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].N = N;
        jobs[t].window = window;
        jobs[t].stream = stream;
        jobs[t].first = frames * t / nthreads;
        jobs[t].count = frames * (t + 1) / nthreads - jobs[t].first;
        jobs[t].pwr = partial + (size_t)t * (size_t)N;
        pthread_create(&th[t], NULL, mt_run, &jobs[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int t = 0; t < nthreads; ++t) s += partial[(size_t)t * (size_t)N + i];
        pwr_out[i] = s;
    }
    if (repeats_done_out) *repeats_done_out = frames;
    free(jobs); free(th); free(partial);
    return rc;
}

/* The all-cores TIMING leg of bench.py's cpu_baseline: like rpf_oracle_accumulate_mt, but every thread keeps
 * ONE worker (its plan and long-double twiddle tables are built once, outside what is worth timing) and walks
 * its frame range `loops` times, so that thread start-up and planning do not dominate the figure printed
 * beside the GPU's.  pwr_out = the sum over one walk (as rpf_oracle_accumulate_mt). */
typedef struct {
    rpf_oracle_worker* w;
    const uint8_t* stream;
    int N;
    int64_t first, count;
    int loops;
    double* pwr;
    int rc;
} mtl_job;

static void* mtl_run(void* arg)
{
    mtl_job* j = (mtl_job*)arg;
    for (int l = 0; l < j->loops; ++l) {
        rpf_oracle_worker_begin(j->w, j->count);
        rpf_oracle_worker_consume(j->w, j->stream + 2 * (size_t)j->N * (size_t)j->first, 2 * (size_t)j->N * (size_t)j->count);
        if (rpf_oracle_worker_repeats_done(j->w) != j->count) j->rc = -2;
    }
    const double* p = rpf_oracle_worker_pwr(j->w);
    for (int i = 0; i < j->N; ++i) j->pwr[i] = p[i];
    return NULL;
}

int rpf_oracle_accumulate_mt_loops(int N, const float* window, const uint8_t* stream, size_t nbytes,
                                   int64_t repeats, int nthreads, int loops, double* pwr_out,
                                   int64_t* repeats_done_out, double* seconds_in_threads)
{
    if (N < 2 || nthreads < 1 || loops < 1) return -1;
    int64_t frames = (int64_t)(nbytes / (2 * (size_t)N));
    if (frames > repeats) frames = repeats;
    if (nthreads > frames) nthreads = frames > 0 ? (int)frames : 1;
    mtl_job* jobs = (mtl_job*)calloc((size_t)nthreads, sizeof(mtl_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    double* partial = (double*)calloc((size_t)nthreads * (size_t)N, sizeof(double));
    if (!jobs || !th || !partial) { free(jobs); free(th); free(partial); return -1; }
    int rc = 0;
    for (int t = 0; t < nthreads; ++t) {          /* planning: not timed */
        jobs[t].w = rpf_oracle_worker_create(N, window, 32);
        if (!jobs[t].w) rc = -1;
        jobs[t].stream = stream;
        jobs[t].N = N;
        jobs[t].first = frames * t / nthreads;
        jobs[t].count = frames * (t + 1) / nthreads - jobs[t].first;
        jobs[t].loops = loops;
        jobs[t].pwr = partial + (size_t)t * (size_t)N;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (rc == 0) {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, mtl_run, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) {
            pthread_join(th[t], NULL);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_in_threads) *seconds_in_threads = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    for (int i = 0; i < N; ++i) {
        double s = 0;
        for (int t = 0; t < nthreads; ++t) s += partial[(size_t)t * (size_t)N + i];
        pwr_out[i] = s;
    }
    for (int t = 0; t < nthreads; ++t)
        if (jobs[t].w) rpf_oracle_worker_destroy(jobs[t].w);
    if (repeats_done_out) *repeats_done_out = frames;
    free(jobs); free(th); free(partial);
    return rc;
}

/* --------------------------------------------------------------- output -- */

static double output_value(const double* pwr, int i, int N, int64_t repeats_done, int samplerate,
                           int linear, const double* baseline)
{
    /* acquisition.cxx:392-399: successive divisions, in this order */
    const double p = pwr[i] / (double)repeats_done / (double)N / (double)samplerate;
    if (linear) return p - (baseline ? baseline[i] : 0);
    return 10 * log10(p) - (baseline ? baseline[i] : 0);
}

long rpf_oracle_format_text(double* pwr, int N, int64_t repeats_done, int64_t tuned_freq,
                            int samplerate, int linear, const double* baseline, char* out,
                            size_t cap)
{
    /* :377 interpolate the central point to cancel the DC bias (in place) */
    pwr[N / 2] = (pwr[N / 2 - 1] + pwr[N / 2 + 1]) / 2;
    /* :380-383 -- note actual_samplerate/params.N is an INTEGER division there */
    const int extraDigitsFreq = 2;
    const int significantPlacesFreq =
        (int)ceil(floor(log10((double)tuned_freq)) - log10((double)(samplerate / N)) + 1 +
                  extraDigitsFreq);
    const int significantPlacesPwr = 6;
    size_t used = 0;
    for (int i = 0; i < N; ++i) {
        /* :391 */
        const double freq = (double)tuned_freq + (i - N / 2.0) * samplerate / N;
        const double v = output_value(pwr, i, N, repeats_done, samplerate, linear, baseline);
        /* :412-417 default-floatfield ostream output == printf %.{prec}g */
        int n = snprintf(out + used, used < cap ? cap - used : 0, "%.*g %.*g\n",
                         significantPlacesFreq, freq, significantPlacesPwr, v);
        if (n < 0 || used + (size_t)n >= cap) return -1;
        used += (size_t)n;
    }
    /* :428-432 blank line after each spectrum */
    if (used + 2 > cap) return -1;
    out[used++] = '\n';
    out[used] = '\0';
    return (long)used;
}

void rpf_oracle_format_matrix(double* pwr, int N, int64_t repeats_done, int samplerate, int linear,
                              const double* baseline, float* row_out)
{
    pwr[N / 2] = (pwr[N / 2 - 1] + pwr[N / 2 + 1]) / 2;                  /* :377 */
    for (int i = 0; i < N; ++i)                                          /* :400-405 */
        row_out[i] = (float)output_value(pwr, i, N, repeats_done, samplerate, linear, baseline);
}

/* ----------------------------------------------------------------- plan -- */

#define RPF_BASE_BUF 16384               /* params.h:26 */
#define RPF_DEFAULT_BUF_MULTIPLIER 100   /* params.h:27 */

int rpf_oracle_make_plan(rpf_oracle_plan_params* p, int64_t* freqs, int cap)
{
    /* acquisition.cxx:162-163 */
    if (p->integration_time_isSet)
        p->repeats = (int64_t)ceil(p->sample_rate * p->integration_time / p->N);
    /* :166-176 */
    if (!p->buf_length_isSet) {
        int64_t base_buf_multiplier = (int64_t)ceil((2.0 * p->N * p->repeats) / RPF_BASE_BUF);
        if (base_buf_multiplier <= RPF_DEFAULT_BUF_MULTIPLIER)
            p->buf_length =
                (int)(RPF_BASE_BUF * ((base_buf_multiplier == 0) ? 1 : base_buf_multiplier));
    }
    /* :180-197 */
    int n = 0;
    if (p->freq_hopping_isSet) {
        double min_overhang = p->sample_rate * p->min_overlap / 100;
        int hops = (int)ceil(((double)(p->stopfreq - p->startfreq) - min_overhang) /
                             ((double)p->sample_rate - min_overhang));
        if (hops > 1) {
            int overhang = (int)(((int64_t)hops * p->sample_rate - (p->stopfreq - p->startfreq)) /
                                 (hops - 1));
            if (cap < hops) return -1;
            freqs[n++] = (int64_t)(p->startfreq + p->sample_rate / 2.0);
            for (int hop = 1; hop < hops; ++hop) {
                freqs[n] = freqs[n - 1] + p->sample_rate - overhang;
                ++n;
            }
        } else {
            if (cap < 1) return -1;
            freqs[n++] = (p->startfreq + p->stopfreq) / 2;
        }
    } else {
        if (cap < 1) return -1;
        freqs[n++] = p->cfreq;
    }
    return n;
}

int64_t rpf_oracle_data_needed(int64_t dataTotal, int64_t dataRead, int buf_length)
{
    /* acquisition.cxx:288-300 */
    int64_t dataNeeded = dataTotal - dataRead;
    if (dataNeeded >= buf_length)
        dataNeeded = buf_length;
    else {
        dataNeeded = (int64_t)(RPF_BASE_BUF * ceil((double)dataNeeded / RPF_BASE_BUF));
        if (dataNeeded > buf_length) dataNeeded = buf_length;
    }
    return dataNeeded;
}
