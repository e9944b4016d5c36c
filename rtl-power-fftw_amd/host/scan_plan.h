// scan_plan.h -- measurement plan (/root/reference/src/acquisition.cxx:158-208):
// repeats from the integration time, automatic buffer length, frequency hops.
#ifndef RPF_HOST_SCAN_PLAN_H
#define RPF_HOST_SCAN_PLAN_H

#include <cstdint>
#include <list>

#include "options.h"

namespace rpf_host {

class Plan {
public:
    // Adjusts options.repeats / options.buf_length like the reference's Plan does.
    Plan(Options& options, int actual_samplerate);
    void print() const;
    std::list<int64_t> freqs_to_tune;
    int actual_samplerate;
private:
    Options& options_;
};

// Bytes the producer asks the source for next (acquisition.cxx:288-300).
int64_t next_read_size(int64_t data_total, int64_t data_read, int buf_length);

}  // namespace rpf_host
#endif
