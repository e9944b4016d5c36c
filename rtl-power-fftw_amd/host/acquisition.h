// acquisition.h -- one data acquisition at one frequency: the producer loop of
// Acquisition::run (/root/reference/src/acquisition.cxx:222-348) over the
// engine's hand-off calls, the stderr summary (:350-358) and the spectrum
// writer (:360-433; text for gnuplot or float32 matrix rows).
#ifndef RPF_HOST_ACQUISITION_H
#define RPF_HOST_ACQUISITION_H

#include <cstdint>
#include <ctime>
#include <exception>
#include <ostream>
#include <string>

#include "aux_data.h"
#include "datastore.h"
#include "options.h"
#include "sample_source.h"

namespace rpf_host {

// Matrix-mode bookkeeping that the reference keeps in globals
// (/root/reference/src/metadata.h:28-33, rtl_power_fftw.cxx:39-48).
struct ScanMetadata {
    int metaRows = 1;
    int metaCols = 0;
    float avgScanDur = 0.0f;
    float sumScanDur = 0.0f;
    time_t scanEnd = 0, scanBeg = 0;
    int tunfreq = 0;
    int startFreq = 0, endFreq = 0, stepFreq = 0;
    std::string firstAcqTimestamp, lastAcqTimestamp;
    int cntTimeStamps = 0;
};

// acquisition.h:62-76 of the reference: tuning failed after three attempts.
class TuneError : public std::exception {
public:
    explicit TuneError(int64_t freq_) : freq(freq_) {}
    const char* what() const noexcept override { return "Could not tune to the given frequency."; }
    int64_t frequency() const { return freq; }
private:
    int64_t freq;
};

// One device's share of an acquisition in a multi-device scan:
// `repeats` frames starting `first_frame` frames into the hop whose bytes begin
// `hop_base` bytes into a sequential replay.
struct Shard {
    int64_t repeats = 0;
    int64_t first_frame = 0;
    uint64_t hop_base = 0;
};

class Acquisition {
public:
    Acquisition(const Options& options, AuxData& aux, SampleSource& source, Datastore& data,
                ScanMetadata& meta, int actual_samplerate, int64_t freq);
    // The same for a shard: quiet (the scan's main thread reports in hop order) and
    // without touching the scan-wide metadata.
    Acquisition(const Options& options, AuxData& aux, SampleSource& source, Datastore& data,
                ScanMetadata& meta, int actual_samplerate, int64_t freq, const Shard& shard);
    void run();
    void print_summary() const;
    void write_data(std::ostream& out) const;

    int64_t tuned_freq() const { return tuned_freq_; }
    const std::string& start_stamp() const { return start_stamp_; }
    const std::string& end_stamp() const { return end_stamp_; }
    int64_t device_readouts() const { return device_readouts_; }
    int64_t successful_readouts() const { return successful_readouts_; }
    static std::string utc_now();

private:
    bool chatty() const { return !sharded_ && (!options_.talkless || options_.outcnt == 0); }

    const Options& options_;
    AuxData& aux_;
    SampleSource& source_;
    Datastore& data_;
    ScanMetadata& meta_;
    int actual_samplerate_;
    int64_t freq_;
    Shard shard_;
    bool sharded_ = false;
    int64_t tuned_freq_ = 0;
    std::string start_stamp_, end_stamp_;
    int64_t device_readouts_ = 0;
    int64_t successful_readouts_ = 0;
};

// stderr summary of one acquisition (acquisition.cxx:350-358)
void print_acquisition_summary(int N, int64_t repeats_done, int64_t device_readouts, int64_t successful_readouts,
                               int actual_samplerate);
// The five '#' lines that precede a text spectrum (acquisition.cxx:411-419)
void write_text_header(std::ostream& out, const std::string& start_stamp, const std::string& end_stamp);

// Text/matrix rendering of one accumulated spectrum (acquisition.cxx:377-432).
// Mutates pwr[N/2] (DC interpolation) exactly like the reference.
void write_spectrum_text(std::ostream& out, std::vector<double>& pwr, int N, int64_t repeats_done,
                         int64_t tuned_freq, int samplerate, bool linear, const std::vector<double>* baseline);
void spectrum_matrix_row(std::vector<double>& pwr, int N, int64_t repeats_done, int samplerate, bool linear,
                         const std::vector<double>* baseline, std::vector<float>& row);
// Matrix mode: append one float32 row to options.bin_file and keep the row/column
// bookkeeping of the .met file (acquisition.cxx:385-388,400-409,421-426).
void append_matrix_row(const Options& options, ScanMetadata& meta, std::vector<double>& pwr, int64_t repeats_done,
                       int64_t tuned_freq, int samplerate, const std::vector<double>* baseline);

}  // namespace rpf_host
#endif
