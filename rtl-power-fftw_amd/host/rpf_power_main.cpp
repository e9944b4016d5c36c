// rpf_power -- the rtl_power_fftw program (/root/reference/src/rtl_power_fftw.cxx:50-233)
// on top of the MI355X engine: same command line, same stderr chatter, same
// gnuplot-compatible stdout / binary-matrix output, same exit codes.  The dongle
// is replaced by a replayed or synthetic byte stream (sample_source.h).
//
// One device (--gpu k, the default): the reference's loop, one acquisition after
// the other (rtl_power_fftw.cxx:132-205).
// Several devices (--gpus a,b,...): single process, one engine, one
// sample source and one producer thread per device.  The hops of a pass are dealt
// hop-major to the devices in contiguous frame-aligned ranges (frame k of the pass
// -> device floor(k G / (hops R)): whole hops when G divides the hop count, frame
// ranges of a hop otherwise); the main thread adds a hop's per-device accumulators
// in device order (fixed order: reproducible) and writes the spectra in hop order,
// so stdout is what one device would have printed.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

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

void note_scan_extent(const Options& options, const Plan& plan, int actual_samplerate, ScanMetadata& meta)
{
    meta.tunfreq = static_cast<int>(plan.freqs_to_tune.front());
    meta.startFreq = static_cast<int>(meta.tunfreq + (0 - options.N / 2.0) * actual_samplerate / options.N);
    meta.tunfreq = static_cast<int>(plan.freqs_to_tune.back());
    meta.endFreq = static_cast<int>(meta.tunfreq + ((options.N - 1) - options.N / 2.0) * actual_samplerate / options.N);
    meta.stepFreq = actual_samplerate / options.N;
}

// ---- several devices --------------------------------------------------------------------
struct DeviceJob {
    int hop = 0;        // index into the pass's hop list
    int part = 0;       // index into that hop's parts
    Shard shard;
};

struct HopPart {
    std::vector<double> pwr;
    int64_t repeats_done = 0, device_readouts = 0, successful_readouts = 0, tuned_freq = 0;
    std::string start_stamp, end_stamp;
    int device_slot = 0;
    bool done = false;
};

// Frame ranges of one pass (hops x repeats frames) for device slot d of G -- the same
// partition as rtl-power-fftw_amd/sharding.py: shard_hops().
std::vector<DeviceJob> device_jobs(int hops, int64_t repeats, int G, int d, std::vector<int>& parts_per_hop)
{
    const int64_t total = static_cast<int64_t>(hops) * repeats;
    int64_t pos = (total * d + G - 1) / G;
    const int64_t end = (total * (d + 1) + G - 1) / G;
    std::vector<DeviceJob> jobs;
    while (pos < end) {
        DeviceJob job;
        job.hop = static_cast<int>(pos / repeats);
        job.shard.first_frame = pos - job.hop * repeats;
        job.shard.repeats = std::min(repeats - job.shard.first_frame, end - pos);
        job.part = parts_per_hop[job.hop]++;
        jobs.push_back(job);
        pos += job.shard.repeats;
    }
    return jobs;
}

// Bytes one hop takes out of a sequential replay (the producer's reads are rounded up to
// whole 16384-byte transfers, acquisition.cxx:288-300).
uint64_t replay_bytes_per_hop(const Options& options)
{
    const int64_t total = 2 * static_cast<int64_t>(options.N) * options.repeats;
    int64_t read = 0;
    while (read < total) read += next_read_size(total, read, options.buf_length);
    return static_cast<uint64_t>(read);
}

class MultiDeviceScan {
public:
    MultiDeviceScan(Options& options, AuxData& aux, SampleSource& source, int actual_samplerate)
        : options_(options), aux_(aux), rate_(actual_samplerate)
    {
        for (size_t d = 0; d < options.devices.size(); ++d) {
            std::unique_ptr<SampleSource> s = source.clone();
            if (!s)
                throw RPFexception("--gpus with several devices needs a source every device can read on its own: "
                                   "--synthetic <seed> or a seekable --input <file> (not stdin, not a dongle).",
                                   ReturnValue::InvalidArgument);
            sources_.push_back(std::move(s));
            stores_.emplace_back(new Datastore(options, aux.window_values, options.devices[d]));
        }
        bytes_per_hop_ = replay_bytes_per_hop(options);
    }
    ~MultiDeviceScan() { rpf_scan_reducer_destroy(reducer_); }

    // north_star / SURVEY.md 8e: the per-device accumulators of a pass meet in ONE RCCL reduce onto the
    // first device (rpf_scan_reducer_*: ncclCommInitAll over the listed devices, librccl loaded with
    // dlopen).  Without RCCL, or with a device list it refuses (the same device twice), the main thread
    // adds them itself in device order -- the same sums up to the order of the additions.
    void prepare_reduce(int max_hops)
    {
        if (options_.reduce == "host" || reducer_ || reducer_tried_) return;
        reducer_tried_ = true;
        int rc = rpf_scan_reducer_create(options_.devices.data(), static_cast<int>(options_.devices.size()), options_.N,
                                         max_hops, &reducer_);
        if (rc != RPF_OK) {
            reducer_ = nullptr;
            if (options_.reduce == "rccl")
                throw RPFexception(std::string("--reduce rccl: ") + rpf_scan_reducer_last_error(nullptr), (ReturnValue)rc);
            if (chatty(options_))
                std::cerr << "RCCL reduce not available (" << rpf_scan_reducer_last_error(nullptr)
                          << "); adding the per-device spectra on the host." << std::endl;
        } else if (chatty(options_)) {
            std::cerr << "Per-device spectra are reduced over RCCL onto gpu " << options_.devices.front() << "." << std::endl;
        }
        reducer_hops_ = max_hops;
    }

    // One pass over `freqs`; spectra to stdout / the matrix file in hop order.  Returns
    // false when the pass was cut short (interrupt, exhausted replay).
    bool run_pass(const std::vector<int64_t>& freqs, ScanMetadata& meta, bool& meta_pending, const Plan& plan,
                  int64_t& last_repeats_done)
    {
        const int H = static_cast<int>(freqs.size()), G = static_cast<int>(stores_.size());
        std::vector<int> parts_per_hop(H, 0);
        std::vector<std::vector<DeviceJob>> jobs(G);
        for (int d = 0; d < G; ++d) jobs[d] = device_jobs(H, options_.repeats, G, d, parts_per_hop);
        std::vector<std::vector<HopPart>> parts(H);
        for (int h = 0; h < H; ++h) parts[h].resize(parts_per_hop[h]);
        for (int d = 0; d < G; ++d)
            for (DeviceJob& j : jobs[d]) {
                j.shard.hop_base = pass_base_ + static_cast<uint64_t>(j.hop) * bytes_per_hop_;
                parts[j.hop][j.part].device_slot = d;
            }

        prepare_reduce(H);
        const bool rccl = reducer_ && H <= reducer_hops_;
        if (rccl && rpf_scan_reducer_begin(reducer_) != RPF_OK)
            throw RPFexception(rpf_scan_reducer_last_error(reducer_), ReturnValue::HardwareError);
        std::vector<double> reduced;                 // rccl: hops x N, filled once every worker has finished

        std::mutex mutex;
        std::condition_variable progress;
        std::string error;
        ReturnValue error_code = ReturnValue::Success;
        std::vector<std::thread> workers;
        std::atomic<bool> cancel(false);       // a worker failed or the pass was cut short: start no further job
        // whatever leaves this scope -- the return at the end or an exception in between -- joins the workers first
        struct JoinAll {
            std::vector<std::thread>& threads;
            std::atomic<bool>& cancel;
            ~JoinAll()
            {
                cancel.store(true);
                for (std::thread& w : threads)
                    if (w.joinable()) w.join();
            }
        } join_all{workers, cancel};
        finished_workers_ = 0;
        for (int d = 0; d < G; ++d)
            workers.emplace_back([&, d]() {
                ScanMetadata unused;
                try {
                    for (const DeviceJob& j : jobs[d]) {
                        if (cancel.load()) break;
                        Acquisition acq(options_, aux_, *sources_[d], *stores_[d], unused, rate_, freqs[j.hop], j.shard);
                        acq.run();
                        if (rccl && stores_[d]->repeats_done > 0 &&
                            rpf_scan_reducer_deposit(reducer_, d, j.hop, stores_[d]->engine()) != RPF_OK)
                            throw RPFexception(rpf_scan_reducer_last_error(reducer_), ReturnValue::HardwareError);
                        std::lock_guard<std::mutex> lock(mutex);
                        HopPart& p = parts[j.hop][j.part];
                        if (!rccl) p.pwr = stores_[d]->pwr;
                        p.repeats_done = stores_[d]->repeats_done;
                        p.device_readouts = acq.device_readouts();
                        p.successful_readouts = acq.successful_readouts();
                        p.tuned_freq = acq.tuned_freq();
                        p.start_stamp = acq.start_stamp();
                        p.end_stamp = acq.end_stamp();
                        p.done = true;
                        progress.notify_all();
                        if (interrupts.load() >= static_cast<int>(InterruptState::FinishNow)) break;
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lock(mutex);
                    if (error.empty()) {
                        error = e.what();
                        const RPFexception* r = dynamic_cast<const RPFexception*>(&e);
                        error_code = r ? r->returnValue() : ReturnValue::AcquisitionError;
                    }
                    cancel.store(true);
                    progress.notify_all();
                }
                std::lock_guard<std::mutex> lock(mutex);
                finished_workers_++;
                progress.notify_all();
            });

        bool complete = true;
        const std::vector<double>* baseline = options_.baseline ? &aux_.baseline_values : nullptr;
        time_t hop_begin = std::time(nullptr);
        if (meta.cntTimeStamps == 0) {
            meta.firstAcqTimestamp = Acquisition::utc_now();
            meta.cntTimeStamps++;
        }
        if (rccl) {
            // the reduce needs every device's block: wait for the workers, then ONE ncclReduce + one D2H
            {
                std::unique_lock<std::mutex> lock(mutex);
                progress.wait(lock, [&]() { return finished_workers_ == G; });
            }
            // a failed worker: nothing was reduced, so nothing may be written (the loop below would
            // otherwise copy hop rows out of the empty `reduced`); JoinAll joins the workers
            if (!error.empty()) throw RPFexception(error, error_code);
            reduced.assign(static_cast<size_t>(H) * options_.N, 0.0);
            if (rpf_scan_reducer_reduce(reducer_, H, reduced.data()) != RPF_OK)
                throw RPFexception(rpf_scan_reducer_last_error(reducer_), ReturnValue::HardwareError);
        }
        for (int h = 0; h < H && complete; ++h) {
            {
                std::unique_lock<std::mutex> lock(mutex);
                auto ready = [&]() {
                    for (const HopPart& p : parts[h])
                        if (!p.done) return false;
                    return true;
                };
                progress.wait(lock, [&]() { return ready() || !error.empty() || finished_workers_ == G; });
                if (!ready()) {
                    complete = false;
                    break;
                }
            }
            // the hop's accumulator: per-device partial sums added in device order
            std::vector<double> pwr(options_.N, 0.0);
            int64_t repeats_done = 0, readouts = 0, successful = 0;
            if (rccl) std::copy(reduced.begin() + static_cast<size_t>(h) * options_.N,
                                reduced.begin() + static_cast<size_t>(h + 1) * options_.N, pwr.begin());
            for (const HopPart& p : parts[h]) {
                if (!rccl)
                    for (int i = 0; i < options_.N; ++i) pwr[i] += p.pwr[i];
                repeats_done += p.repeats_done;
                readouts += p.device_readouts;
                successful += p.successful_readouts;
            }
            const HopPart& first = parts[h].front();
            const HopPart& last = parts[h].back();
            if (chatty(options_)) {
                std::cerr << "Tuning to " << freqs[h] << " Hz (" << parts[h].size() << " device"
                          << (parts[h].size() == 1 ? "" : "s") << ", first: gpu "
                          << options_.devices[first.device_slot] << ")" << std::endl;
                std::cerr << "Device tuned to: " << first.tuned_freq << " Hz" << std::endl;
                std::cerr << "Acquisition started at " << first.start_stamp << std::endl;
                std::cerr << "Acquisition done at " << last.end_stamp << std::endl;
                print_acquisition_summary(options_.N, repeats_done, readouts, successful, rate_);
            }
            time_t now = std::time(nullptr);
            meta.scanBeg = hop_begin;
            meta.scanEnd = now;
            meta.lastAcqTimestamp = Acquisition::utc_now();
            meta.sumScanDur += static_cast<float>(std::difftime(now, hop_begin));
            meta.avgScanDur = meta.sumScanDur / meta.metaRows;
            hop_begin = now;
            last_repeats_done = repeats_done;
            if (repeats_done == 0) {
                std::cerr << "No complete spectrum at " << freqs[h] << " Hz (input exhausted); nothing written."
                          << std::endl;
                complete = false;
                break;
            }
            if (options_.matrixMode && meta_pending) {
                note_scan_extent(options_, plan, rate_, meta);
                meta_pending = false;
            }
            if (!options_.matrixMode) {
                write_text_header(std::cout, first.start_stamp, last.end_stamp);
                write_spectrum_text(std::cout, pwr, options_.N, repeats_done, first.tuned_freq, rate_,
                                    options_.linear, baseline);
            } else {
                append_matrix_row(options_, meta, pwr, repeats_done, first.tuned_freq, rate_, baseline);
            }
            hops_written_++;
            if (chatty(options_))
                for (const HopPart& p : parts[h]) {
                    std::cerr << "gpu " << options_.devices[p.device_slot] << ": ";
                    stores_[p.device_slot]->printQueueHistogram();
                }
            if (checkInterrupt(InterruptState::FinishNow)) complete = false;
        }
        if (!complete) cancel.store(true);
        for (std::thread& w : workers) w.join();
        if (!error.empty()) throw RPFexception(error, error_code);
        pass_base_ += static_cast<uint64_t>(H) * bytes_per_hop_;
        for (const auto& s : sources_)
            if (s->exhausted()) exhausted_ = true;
        return complete;
    }

    bool exhausted() const { return exhausted_; }
    int64_t hops_written() const { return hops_written_; }

private:
    Options& options_;
    AuxData& aux_;
    int rate_;
    std::vector<std::unique_ptr<SampleSource>> sources_;
    std::vector<std::unique_ptr<Datastore>> stores_;
    uint64_t bytes_per_hop_ = 0, pass_base_ = 0;
    int finished_workers_ = 0;
    rpf_scan_reducer* reducer_ = nullptr;
    bool reducer_tried_ = false;
    int reducer_hops_ = 0;
    bool exhausted_ = false;
    int64_t hops_written_ = 0;
};

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

    const bool multi = options.devices.size() > 1;
    std::unique_ptr<Datastore> data;                 // after Plan fixed N / repeats / buf_length
    std::unique_ptr<MultiDeviceScan> scan;
    if (multi) {
        scan.reset(new MultiDeviceScan(options, aux, *source, actual_samplerate));
        std::cerr << "Scan spread over " << options.devices.size() << " engines (gpus";
        for (int d : options.devices) std::cerr << " " << d;
        std::cerr << ")" << std::endl;
    } else {
        data.reset(new Datastore(options, aux.window_values));
    }
    set_CtrlC_handler(true);
    if (options.session_duration_isSet) exit_time += std::time(nullptr);
    if (options.matrixMode) std::ofstream(options.bin_file, std::ios::out | std::ios::trunc | std::ios::binary);

    ScanMetadata meta;
    bool meta_pending = true;
    options.finalfreq = static_cast<int>(plan.freqs_to_tune.back());
    int64_t last_repeats_done = 0;
    bool wrote_anything = false;
    bool stop = false;
    do {
        bool input_ended = false;
        if (multi) {
            const std::vector<int64_t> freqs(plan.freqs_to_tune.begin(), plan.freqs_to_tune.end());
            const bool complete = scan->run_pass(freqs, meta, meta_pending, plan, last_repeats_done);
            wrote_anything = scan->hops_written() > 0;
            input_ended = scan->exhausted();
            if (!complete && input_ended) stop = true;
        } else {
            for (auto hop = plan.freqs_to_tune.begin(); hop != plan.freqs_to_tune.end();) {
                Acquisition acquisition(options, aux, *source, *data, meta, actual_samplerate, *hop);
                try {
                    acquisition.run();
                    ++hop;
                } catch (TuneError& e) {
                    std::cerr << "Unable to tune to " << e.frequency() << ". Dropping from frequency list." << std::endl;
                    hop = plan.freqs_to_tune.erase(hop);
                    continue;
                }
                if (chatty(options)) acquisition.print_summary();
                last_repeats_done = data->repeats_done;
                if (data->repeats_done == 0) {
                    // the reference would divide by zero here and print a spectrum of NaNs (acquisition.cxx:393)
                    std::cerr << "No complete spectrum at " << acquisition.tuned_freq()
                              << " Hz; nothing written." << std::endl;
                    // (a second Ctrl-C must end the scan here too, not retune once per remaining hop)
                    if (source->exhausted() || checkInterrupt(InterruptState::FinishNow)) break;
                    continue;
                }
                if (options.matrixMode && meta_pending) {
                    note_scan_extent(options, plan, actual_samplerate, meta);
                    meta_pending = false;
                }
                acquisition.write_data(std::cout);
                wrote_anything = true;
                if (chatty(options)) data->printQueueHistogram();
                if (source->exhausted()) break;
                if (checkInterrupt(InterruptState::FinishNow)) break;
            }
            input_ended = source->exhausted();
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
        if (input_ended) {
            // a finite replay has nothing more to give: --continue / -e would spin on empty reads
            if (!stop) std::cerr << "Input exhausted, ending the session." << std::endl;
            stop = true;
        }
    } while (!stop);

    if (options.matrixMode) write_metadata(options, meta, last_repeats_done, actual_samplerate);
    if (plan.freqs_to_tune.empty())
        throw RPFexception("No valid frequencies left.", ReturnValue::AcquisitionError);
    if (!wrote_anything)
        throw RPFexception("No complete spectrum could be acquired (input too short?).", ReturnValue::AcquisitionError);
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
