// options.h -- the rtl_power_fftw command line (/root/reference/src/params.h:33-66,
// params.cxx:101-272; doc/rtl_power_fftw.1.md:21-87): the same 19 options with the
// same defaults, derived values, validation messages and exit codes, parsed by a
// small table-driven parser (TCLAP is not available here).  Additive options
// select the sample source, since there is no dongle next to an MI355X:
//   --input <file|->      replay interleaved u8 IQ from a file / stdin
//   --synthetic <seed>    built-in receiver-like generator
//   (neither: a live RTL-SDR dongle through librtlsdr, like the reference)
//   --gpu <ordinal>       HIP device
//   --gpus <a,b,...>      several HIP devices: the hops of a scan (and, with fewer hops than
//                         devices, frame ranges of a hop) are dealt to one engine per device
#ifndef RPF_HOST_OPTIONS_H
#define RPF_HOST_OPTIONS_H

#include <cstdint>
#include <string>
#include <vector>

#include "datastore.h"

namespace rpf_host {

const int base_buf = 16384;                // params.h:26
const int default_buf_multiplier = 100;    // params.h:27

struct Options : Params {
    // everything the hot path needs lives in Params (datastore.h); the rest:
    int dev_index = 0;
    int gain = 372;
    int64_t startfreq = 0;
    int64_t stopfreq = 0;
    double integration_time = 0;
    bool integration_time_isSet = false;
    bool buf_length_isSet = false;
    double min_overlap = 0;
    int ppm_error = 0;
    bool endless = false;
    bool strict_time = false;
    std::string baseline_file;
    std::string window_file;
    bool freq_hopping_isSet = false;
    int outcnt = 0;
    double session_duration = 0;
    bool session_duration_isSet = false;
    bool talkless = false;
    bool matrixMode = false;
    int finalfreq = 0;
    std::string matrix_file, bin_file, meta_file;
    // additive
    std::string input_file;          // --input
    bool synthetic = false;          // --synthetic given
    uint64_t synthetic_seed = 2;
    std::vector<int> devices;        // --gpus a,b,... (else the single --gpu ordinal)
    std::string reduce;              // --reduce rccl|host ("" = rccl if it loads, else host)
    bool show_help = false, show_version = false;
};

// Throws RPFexception(InvalidArgument / TCLAPerror) like Params::Params does.
Options parse_command_line(int argc, const char* const* argv);
std::string usage_text();
extern const char* const kVersion;

}  // namespace rpf_host
#endif
