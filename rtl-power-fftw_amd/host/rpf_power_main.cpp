// rpf_power -- the rtl_power_fftw program (/root/reference/src/rtl_power_fftw.cxx:50-233)
// on top of the MI355X engine: same command line, same stderr chatter, same
// gnuplot-compatible stdout / binary-matrix output, same exit codes.  The dongle
// is replaced by a replayed or synthetic byte stream (sample_source.h).
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>

#include "acquisition.h"
#include "aux_data.h"
#include "datastore.h"
#include "interrupts.h"
#include "options.h"
#include "sample_source.h"
#include "scan_plan.h"

using namespace rpf_host;

namespace {

bool chatty(const Options& o) { return !o.talkless || o.outcnt == 0; }

void write_metadata(const Options& o, const ScanMetadata& m, int64_t repeats_done, int samplerate)
{
    // rtl_power_fftw.cxx:207-220; example in doc/rtl_power_fftw.1.md:186-194
    std::ofstream meta(o.meta_file, std::ios::out | std::ios::trunc);
    meta << m.metaCols << " # frequency bins (columns)" << std::endl;
    meta << (m.metaRows - 1) << " # scans (rows)" << std::endl;
    meta << m.startFreq << " # startFreq (Hz)" << std::endl;
    meta << m.endFreq << " # endFreq (Hz)" << std::endl;
    meta << m.stepFreq << " # stepFreq (Hz)" << std::endl;
    meta << static_cast<double>(o.N) * repeats_done / samplerate << " # effective integration time secs" << std::endl;
    meta << m.avgScanDur << " # avgScanDur (sec)" << std::endl;
    meta << m.firstAcqTimestamp << " # firstAcqTimestamp UTC" << std::endl;
    meta << m.lastAcqTimestamp << " # lastAcqTimestamp UTC" << std::endl;
}

int run(int argc, char** argv)
{
    Options options = parse_command_line(argc, argv);
    if (options.show_help) {
        std::cout << usage_text();
        return 0;
    }
    if (options.show_version) {
        std::cout << argv[0] << "  version: " << kVersion << std::endl;
        return 0;
    }
    AuxData aux(options);

    std::unique_ptr<SampleSource> source;
    RtlSdrSource* dongle = nullptr;
    if (!options.input_file.empty()) source.reset(new FileSource(options.input_file));
    else if (options.synthetic) source.reset(new SyntheticSource(options.synthetic_seed));
    else source.reset(dongle = new RtlSdrSource(options.dev_index));   // rtl_power_fftw.cxx:65

    if (options.endless) options.session_duration_isSet = false;
    time_t exit_time = 0;
    if (options.session_duration_isSet) {
        exit_time = static_cast<int>(options.session_duration);
        std::cerr << "Scan session duration: " << exit_time << " seconds" << std::endl;
    }
    if (dongle) {
        // rtl_power_fftw.cxx:77-97: nearest available gain, provisional tuning, ppm
        dongle->print_gains();
        const int gain = dongle->nearest_gain(options.gain);
        std::cerr << "Selected nearest available gain: " << gain << " (" << 0.1 * gain << " dB)" << std::endl;
        dongle->set_gain(gain);
        try {
            dongle->set_frequency(options.cfreq);
        } catch (RPFexception&) {
        }
        if (options.ppm_error != 0) {
            dongle->set_freq_correction(options.ppm_error);
            std::cerr << "PPM error set to: " << options.ppm_error << std::endl;
        }
    } else {
        source->set_frequency(options.cfreq);
    }
    source->set_sample_rate(static_cast<uint32_t>(options.sample_rate));
    const int actual_samplerate = source->sample_rate();
    std::cerr << "Actual sample rate: " << actual_samplerate << " Hz" << std::endl;

    Plan plan(options, actual_samplerate);
    plan.print();

    Datastore data(options, aux.window_values);      // after Plan fixed N / repeats / buf_length
    set_CtrlC_handler(true);
    if (options.session_duration_isSet) exit_time += std::time(nullptr);
    if (options.matrixMode) std::ofstream(options.bin_file, std::ios::out | std::ios::trunc | std::ios::binary);

    ScanMetadata meta;
    bool meta_pending = true;
    options.finalfreq = static_cast<int>(plan.freqs_to_tune.back());
    bool stop = false;
    do {
        for (auto hop = plan.freqs_to_tune.begin(); hop != plan.freqs_to_tune.end();) {
            Acquisition acquisition(options, aux, *source, data, meta, actual_samplerate, *hop);
            try {
                acquisition.run();
                ++hop;
            } catch (TuneError& e) {
                std::cerr << "Unable to tune to " << e.frequency() << ". Dropping from frequency list." << std::endl;
                hop = plan.freqs_to_tune.erase(hop);
                continue;
            }
            if (chatty(options)) acquisition.print_summary();
            if (options.matrixMode && meta_pending) {
                meta.tunfreq = static_cast<int>(plan.freqs_to_tune.front());
                meta.startFreq = static_cast<int>(meta.tunfreq + (0 - options.N / 2.0) * actual_samplerate / options.N);
                meta.tunfreq = static_cast<int>(plan.freqs_to_tune.back());
                meta.endFreq = static_cast<int>(meta.tunfreq + ((options.N - 1) - options.N / 2.0) * actual_samplerate / options.N);
                meta.stepFreq = actual_samplerate / options.N;
                meta_pending = false;
            }
            acquisition.write_data(std::cout);
            if (chatty(options)) data.printQueueHistogram();
            if (checkInterrupt(InterruptState::FinishNow)) break;
        }
        if (options.talkless && options.outcnt == 0) options.outcnt++;

        // a second blank line closes a full pass over the hops (man page, FREQUENCY SCANNING)
        if (options.session_duration_isSet) {
            if (std::time(nullptr) >= exit_time) {
                stop = true;
                std::cerr << "Session duration elapsed." << std::endl;
                std::cout << std::endl;
            }
        } else {
            std::cout << std::endl;
        }
        if (options.endless) stop = false;
        if (!options.session_duration_isSet && !options.endless) stop = true;
        if (checkInterrupt(InterruptState::FinishPass)) stop = true;
        if (plan.freqs_to_tune.empty()) stop = true;
    } while (!stop);

    if (options.matrixMode) write_metadata(options, meta, data.repeats_done, actual_samplerate);
    if (plan.freqs_to_tune.empty())
        throw RPFexception("No valid frequencies left.", ReturnValue::AcquisitionError);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        return run(argc, argv);
    } catch (RPFexception& e) {
        std::cerr << e.what() << std::endl;
        return static_cast<int>(e.returnValue());
    }
}
