// host_capi.cpp -- TEST SHIM: C entry points onto the CPU-only pieces of the C++
// host (option parsing, unit parsing, Plan, aux-file parsing, spectrum writer)
// so that tests/ can drive them through ctypes and compare with the oracle's
// restatements and the man page.  Not needed by the CLI itself.
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "acquisition.h"
#include "aux_data.h"
#include "options.h"
#include "sample_source.h"
#include "scan_plan.h"
#include "units.h"

using namespace rpf_host;

namespace {
std::string g_text;
int copy_out(const std::string& s, char* out, size_t cap)
{
    if (s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return static_cast<int>(s.size());
}
}  // namespace

extern "C" {

long long rpf_host_parse_frequency(const char* s) { return parse_frequency(s); }
double rpf_host_parse_time(const char* s) { return parse_time(s); }

// Parses argv; on success fills the numeric fields and returns 0, else returns the
// reference's exit code and puts the message in `msg`.
int rpf_host_parse(int argc, const char* const* argv, int* N, int* buffers, int* buf_length,
                   long long* repeats, int* sample_rate, long long* cfreq, long long* startfreq,
                   long long* stopfreq, int* flags, double* integration_time, char* msg, size_t cap)
{
    try {
        Options o = parse_command_line(argc, argv);
        *N = o.N; *buffers = o.buffers; *buf_length = o.buf_length; *repeats = o.repeats;
        *sample_rate = o.sample_rate; *cfreq = o.cfreq; *startfreq = o.startfreq; *stopfreq = o.stopfreq;
        *integration_time = o.integration_time;
        *flags = (o.window ? 1 : 0) | (o.baseline ? 2 : 0) | (o.linear ? 4 : 0) | (o.endless ? 8 : 0) |
                 (o.strict_time ? 16 : 0) | (o.matrixMode ? 32 : 0) | (o.freq_hopping_isSet ? 64 : 0) |
                 (o.talkless ? 128 : 0) | (o.buf_length_isSet ? 256 : 0) | (o.integration_time_isSet ? 512 : 0) |
                 (o.session_duration_isSet ? 1024 : 0) | (o.show_help ? 2048 : 0) | (o.show_version ? 4096 : 0);
        return 0;
    } catch (RPFexception& e) {
        copy_out(e.what(), msg, cap);
        return static_cast<int>(e.returnValue());
    }
}

// Plan on top of a parsed command line: returns hop count (<= cap) or -code.
int rpf_host_plan(int argc, const char* const* argv, int samplerate, long long* repeats, int* buf_length,
                  long long* freqs, int cap)
{
    try {
        Options o = parse_command_line(argc, argv);
        Plan plan(o, samplerate);
        *repeats = o.repeats;
        *buf_length = o.buf_length;
        int n = 0;
        for (auto f : plan.freqs_to_tune) {
            if (n >= cap) return -1;
            freqs[n++] = f;
        }
        return n;
    } catch (RPFexception& e) {
        return -static_cast<int>(e.returnValue());
    }
}

long long rpf_host_next_read_size(long long total, long long done, int buf_length)
{
    return next_read_size(total, done, buf_length);
}

// Aux parser on in-memory text: kind 0 = float column, 1 = double column.
int rpf_host_read_column(const char* text, int kind, double* out, int cap)
{
    std::istringstream in(text);
    int n = 0;
    if (kind == 0) {
        for (float v : read_value_column<float>(in)) { if (n >= cap) return -1; out[n++] = v; }
    } else {
        for (double v : read_value_column<double>(in)) { if (n >= cap) return -1; out[n++] = v; }
    }
    return n;
}

// AuxData with both inputs on "stdin" (text): baseline first, then window.
int rpf_host_aux_from_stdin(int N, int want_window, int want_baseline, const char* text, float* window,
                            double* baseline, char* msg, size_t cap)
{
    Options o;
    o.N = N;
    o.window = want_window != 0;
    o.baseline = want_baseline != 0;
    o.window_file = "-";
    o.baseline_file = "-";
    std::istringstream in(text);
    try {
        AuxData aux(o, in);
        for (size_t i = 0; i < aux.window_values.size(); ++i) window[i] = aux.window_values[i];
        for (size_t i = 0; i < aux.baseline_values.size(); ++i) baseline[i] = aux.baseline_values[i];
        return 0;
    } catch (RPFexception& e) {
        copy_out(e.what(), msg, cap);
        return static_cast<int>(e.returnValue());
    }
}

long rpf_host_format_text(double* pwr, int N, long long repeats_done, long long tuned_freq, int samplerate,
                          int linear, const double* baseline, char* out, size_t cap)
{
    std::vector<double> p(pwr, pwr + N), b;
    if (baseline) b.assign(baseline, baseline + N);
    std::ostringstream os;
    write_spectrum_text(os, p, N, repeats_done, tuned_freq, samplerate, linear != 0, baseline ? &b : nullptr);
    std::memcpy(pwr, p.data(), sizeof(double) * N);
    return copy_out(os.str(), out, cap);
}

void rpf_host_format_matrix(double* pwr, int N, long long repeats_done, int samplerate, int linear,
                            const double* baseline, float* row_out)
{
    std::vector<double> p(pwr, pwr + N), b;
    if (baseline) b.assign(baseline, baseline + N);
    std::vector<float> row;
    spectrum_matrix_row(p, N, repeats_done, samplerate, linear != 0, baseline ? &b : nullptr, row);
    std::memcpy(row_out, row.data(), sizeof(float) * N);
    std::memcpy(pwr, p.data(), sizeof(double) * N);
}

void rpf_host_synthetic(unsigned long long seed, unsigned long long first, unsigned long long n, unsigned char* out)
{
    SyntheticSource::generate(seed, first, n, out);
}

}  // extern "C"
