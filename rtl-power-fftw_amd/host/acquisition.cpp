#include "acquisition.h"

#include <chrono>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "interrupts.h"
#include "scan_plan.h"

namespace rpf_host {

Acquisition::Acquisition(const Options& options, AuxData& aux, SampleSource& source, Datastore& data,
                         ScanMetadata& meta, int actual_samplerate, int64_t freq)
    : options_(options), aux_(aux), source_(source), data_(data), meta_(meta),
      actual_samplerate_(actual_samplerate), freq_(freq)
{
    shard_.repeats = options.repeats;
}

Acquisition::Acquisition(const Options& options, AuxData& aux, SampleSource& source, Datastore& data,
                         ScanMetadata& meta, int actual_samplerate, int64_t freq, const Shard& shard)
    : options_(options), aux_(aux), source_(source), data_(data), meta_(meta),
      actual_samplerate_(actual_samplerate), freq_(freq), shard_(shard), sharded_(true)
{
}

std::string Acquisition::utc_now()
{
    const time_t now = std::time(nullptr);
    char text[80];
    // gmtime_r: several acquisitions run side by side under --gpus a,b,... and std::gmtime hands every caller the same
    // static struct tm (ThreadSanitizer found the race, profiles/r04_tsan.txt)
    struct tm parts;
    gmtime_r(&now, &parts);
    std::strftime(text, sizeof(text), "%Y-%m-%d %X UTC", &parts);
    return text;
}

void Acquisition::run()
{
    // Tune, up to three tries (acquisition.cxx:229-249)
    bool tuned = false;
    for (int attempt = 1; attempt <= 3 && !tuned; ++attempt) {
        if (chatty()) std::cerr << "Tuning to " << freq_ << " Hz (try " << attempt << ")" << std::endl;
        try {
            source_.set_frequency(freq_);
            tuned_freq_ = source_.frequency();
            tuned = tuned_freq_ != 0;
        } catch (const RPFexception&) {
        }
    }
    if (!tuned) throw TuneError(freq_);
    if (chatty()) std::cerr << "Device tuned to: " << tuned_freq_ << " Hz" << std::endl;
    if (sharded_ && !source_.position(shard_.hop_base, 2 * static_cast<uint64_t>(options_.N) *
                                                          static_cast<uint64_t>(shard_.first_frame)))
        throw RPFexception("This sample source cannot be split across devices.", ReturnValue::InvalidArgument);

    data_.begin(shard_.repeats);                                // :252-256

    start_stamp_ = utc_now();
    if (!sharded_) {
        std::time(&meta_.scanBeg);
        if (meta_.cntTimeStamps == 0) {
            meta_.firstAcqTimestamp = utc_now();
            meta_.cntTimeStamps++;
        }
    }
    if (chatty()) std::cerr << "Acquisition started at " << start_stamp_ << std::endl;

    using clock = std::chrono::steady_clock;
    const clock::time_point deadline =
        clock::now() + std::chrono::milliseconds(static_cast<int64_t>(options_.integration_time * 1000));

    const int64_t data_total = 2 * static_cast<int64_t>(options_.N) * shard_.repeats;     // :273
    int64_t data_read = 0;
    while (data_read < data_total) {
        Buffer buffer = data_.acquire();                        // :278-285
        const int64_t wanted = next_read_size(data_total, data_read, options_.buf_length);
        buffer.resize(static_cast<size_t>(wanted));             // :302
        const bool ok = source_.read(buffer);                   // :304
        device_readouts_++;
        if (!ok) {
            std::cerr << "Error: dropped samples." << std::endl;
            data_.unget(buffer);                                // :310-314
            if (!source_.retry_after_short_read()) break;       // a finite replay has nothing more to give
        } else {
            successful_readouts_++;
            data_read += static_cast<int64_t>(buffer.size());   // == wanted, except for the last read of a replay
            data_.submit(buffer);                               // :320-323
            if (source_.exhausted()) break;                     // the replay ended inside this buffer
        }
        if (options_.strict_time && clock::now() >= deadline) break;                      // :326-327
        if (interrupts && checkInterrupt(InterruptState::FinishNow)) break;               // :330-331
    }

    end_stamp_ = utc_now();
    if (!sharded_) {
        std::time(&meta_.scanEnd);
        meta_.lastAcqTimestamp = utc_now();
        meta_.sumScanDur += static_cast<float>(std::difftime(meta_.scanEnd, meta_.scanBeg));
        meta_.avgScanDur = meta_.sumScanDur / meta_.metaRows;
    }
    if (chatty()) std::cerr << "Acquisition done at " << end_stamp_ << std::endl;

    data_.finish();                                             // :343-347
}

void print_acquisition_summary(int N, int64_t repeats_done, int64_t device_readouts, int64_t successful_readouts,
                               int actual_samplerate)
{
    std::cerr << "Actual number of (complex) samples collected: " << static_cast<int64_t>(N) * repeats_done
              << std::endl;
    std::cerr << "Actual number of device readouts: " << device_readouts << std::endl;
    std::cerr << "Number of successful readouts: " << successful_readouts << std::endl;
    std::cerr << "Actual number of averaged spectra: " << repeats_done << std::endl;
    std::cerr << "Effective integration time: " << static_cast<double>(N) * repeats_done / actual_samplerate
              << " seconds" << std::endl;
}

void Acquisition::print_summary() const
{
    print_acquisition_summary(options_.N, data_.repeats_done, device_readouts_, successful_readouts_,
                              actual_samplerate_);
}

void write_text_header(std::ostream& out, const std::string& start_stamp, const std::string& end_stamp)
{
    out << "# rtl-power-fftw output" << std::endl;
    out << "# Acquisition start: " << start_stamp << std::endl;
    out << "# Acquisition end: " << end_stamp << std::endl;
    out << "#" << std::endl;
    out << "# frequency [Hz] power spectral density [dB/Hz]" << std::endl;
}

namespace {

// value of bin i after normalisation (acquisition.cxx:392-399): successive
// divisions in this order, then optional dB and baseline
inline double bin_value(const std::vector<double>& pwr, int i, int N, int64_t repeats_done, int samplerate,
                        bool linear, const std::vector<double>* baseline)
{
    const double p = pwr[i] / repeats_done / N / samplerate;
    const double b = baseline ? (*baseline)[i] : 0;
    return (linear ? p : 10 * std::log10(p)) - b;
}

inline void interpolate_dc(std::vector<double>& pwr, int N)
{
    pwr[N / 2] = (pwr[N / 2 - 1] + pwr[N / 2 + 1]) / 2;         // :377
}

}  // namespace

void write_spectrum_text(std::ostream& out, std::vector<double>& pwr, int N, int64_t repeats_done,
                         int64_t tuned_freq, int samplerate, bool linear, const std::vector<double>* baseline)
{
    interpolate_dc(pwr, N);
    // digits needed to tell neighbouring bins apart, plus two (:380-383; samplerate/N is an int division)
    const int freq_digits = static_cast<int>(
        std::ceil(std::floor(std::log10(static_cast<double>(tuned_freq))) - std::log10(samplerate / N) + 1 + 2));
    for (int i = 0; i < N; ++i) {
        const double freq = tuned_freq + (i - N / 2.0) * samplerate / N;
        out << std::setprecision(freq_digits) << freq << " " << std::setprecision(6)
            << bin_value(pwr, i, N, repeats_done, samplerate, linear, baseline) << std::endl;
    }
    out << std::endl;                                           // :428-432
    out.flush();
}

void spectrum_matrix_row(std::vector<double>& pwr, int N, int64_t repeats_done, int samplerate, bool linear,
                         const std::vector<double>* baseline, std::vector<float>& row)
{
    interpolate_dc(pwr, N);
    row.resize(N);
    for (int i = 0; i < N; ++i)
        row[i] = static_cast<float>(bin_value(pwr, i, N, repeats_done, samplerate, linear, baseline));
}

void Acquisition::write_data(std::ostream& out) const
{
    const std::vector<double>* baseline = options_.baseline ? &aux_.baseline_values : nullptr;
    if (!options_.matrixMode) {
        write_text_header(out, start_stamp_, end_stamp_);
        write_spectrum_text(out, data_.pwr, options_.N, data_.repeats_done, tuned_freq_, actual_samplerate_,
                            options_.linear, baseline);
        return;
    }
    append_matrix_row(options_, meta_, data_.pwr, data_.repeats_done, tuned_freq_, actual_samplerate_, baseline);
}

void append_matrix_row(const Options& options, ScanMetadata& meta, std::vector<double>& pwr, int64_t repeats_done,
                       int64_t tuned_freq, int samplerate, const std::vector<double>* baseline)
{
    std::vector<float> row;
    spectrum_matrix_row(pwr, options.N, repeats_done, samplerate, options.linear, baseline, row);
    std::ofstream bin(options.bin_file, std::ios::out | std::ios::app | std::ios::binary);
    bin.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * 4));
    if (meta.metaRows == 1) meta.metaCols += options.N;
    if (tuned_freq >= options.finalfreq) meta.metaRows++;
}

}  // namespace rpf_host
