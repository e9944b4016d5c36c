#include "scan_plan.h"

#include <cmath>
#include <iostream>

namespace rpf_host {

Plan::Plan(Options& o, int samplerate) : actual_samplerate(samplerate), options_(o)
{
    // -t: as many spectra as the true sample rate delivers in that time
    if (o.integration_time_isSet)
        o.repeats = static_cast<int64_t>(std::ceil(samplerate * o.integration_time / o.N));

    // Short acquisitions get the smallest buffer (a multiple of 16384 bytes) that
    // holds them; anything above ~1.6 MB keeps the 100 x 16384 default.
    if (!o.buf_length_isSet) {
        const int64_t multiples = static_cast<int64_t>(std::ceil((2.0 * o.N * o.repeats) / base_buf));
        if (multiples <= default_buf_multiplier)
            o.buf_length = static_cast<int>(base_buf * (multiples == 0 ? 1 : multiples));
    }

    if (!o.freq_hopping_isSet) {
        freqs_to_tune.push_back(o.cfreq);
        return;
    }
    // Cover [startfreq, stopfreq] exactly with equally spaced, possibly overlapping hops.
    const double span = static_cast<double>(o.stopfreq - o.startfreq);
    const double min_overhang = samplerate * o.min_overlap / 100;
    const int hops = static_cast<int>(std::ceil((span - min_overhang) / (static_cast<double>(samplerate) - min_overhang)));
    if (hops <= 1) {
        freqs_to_tune.push_back((o.startfreq + o.stopfreq) / 2);
        return;
    }
    const int overhang = static_cast<int>((static_cast<int64_t>(hops) * samplerate - (o.stopfreq - o.startfreq)) / (hops - 1));
    int64_t f = static_cast<int64_t>(o.startfreq + samplerate / 2.0);
    for (int hop = 0; hop < hops; ++hop) {
        freqs_to_tune.push_back(f);
        f += samplerate - overhang;
    }
}

void Plan::print() const
{
    const Options& o = options_;
    std::cerr << "Number of bins: " << o.N << std::endl;
    std::cerr << "Total number of (complex) samples to collect: " << static_cast<int64_t>(o.N) * o.repeats << std::endl;
    std::cerr << "Buffer length: " << o.buf_length << std::endl;
    std::cerr << "Number of averaged spectra: " << o.repeats << std::endl;
    std::cerr << "Estimated time of measurements: " << static_cast<double>(o.N) * o.repeats / actual_samplerate
              << " seconds" << std::endl;
    if (o.strict_time)
        std::cerr << "Acquisition will unconditionally terminate after " << o.integration_time << " seconds."
                  << std::endl;
}

int64_t next_read_size(int64_t data_total, int64_t data_read, int buf_length)
{
    int64_t needed = data_total - data_read;
    if (needed >= buf_length) return buf_length;
    // the tail is rounded up to whole 16384-byte USB transfers, capped at one buffer
    needed = static_cast<int64_t>(base_buf * std::ceil(static_cast<double>(needed) / base_buf));
    return needed > buf_length ? buf_length : needed;
}

}  // namespace rpf_host
