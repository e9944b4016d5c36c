// units.h -- number-with-suffix parsing of the rtl_power_fftw command line:
// frequencies take k/M/G (/root/reference/src/params.cxx:29-43), durations take
// d/h/m/s components, each at most once (params.cxx:45-88).
#ifndef RPF_HOST_UNITS_H
#define RPF_HOST_UNITS_H

#include <cstdint>
#include <string>

namespace rpf_host {

// "1420405752", "1420.4M", "1.42G", "144100k" -> Hz; anything else -> -1.
int64_t parse_frequency(const std::string& text);

// "90", "90s", "1h30m", "2d4h10m5s" -> seconds; malformed or a repeated unit -> -1.
double parse_time(const std::string& text);

}  // namespace rpf_host
#endif
