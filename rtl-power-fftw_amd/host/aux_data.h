// aux_data.h -- window function and baseline data read from text
// (/root/reference/src/acquisition.cxx:32-156, doc/rtl_power_fftw.1.md:123-129):
// one value per line, the LAST number of a line counts (so the program's own
// two-column output can be fed back as a baseline), lines starting with '#' are
// comments, and exactly N values must be found (else InvalidInput, exit code 5).
//
// Two defects of the reference are deliberately not reproduced:
//  * "-w file -B file" fails there because one ifstream is reused without
//    close() (acquisition.cxx:108,135); here each file gets its own stream;
//  * "-w - -B -" splits stdin using the size of a still-empty vector
//    (acquisition.cxx:76) so all 2N values land in the baseline; here, as the man
//    page says, the baseline comes first and the window second.
#ifndef RPF_HOST_AUX_DATA_H
#define RPF_HOST_AUX_DATA_H

#include <istream>
#include <vector>

#include "options.h"

namespace rpf_host {

template <typename T>
std::vector<T> read_value_column(std::istream& in);

class AuxData {
public:
    explicit AuxData(const Options& options);
    AuxData(const Options& options, std::istream& standard_input);
    std::vector<double> baseline_values;
    std::vector<float> window_values;
private:
    void load(const Options& options, std::istream& standard_input);
};

}  // namespace rpf_host
#endif
