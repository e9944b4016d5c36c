#include "sample_source.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <thread>

namespace rpf_host {

FileSource::FileSource(const std::string& path) : path_(path)
{
    if (path == "-") {
        file_ = stdin;
    } else {
        file_ = std::fopen(path.c_str(), "rb");
        owns_ = true;
        if (!file_) throw RPFexception("Could not open " + path + ". Quitting.", ReturnValue::InvalidInput);
    }
}

FileSource::~FileSource()
{
    if (file_ && owns_) std::fclose(file_);
}

bool FileSource::read(Buffer& buffer)
{
    // The last read of a replay is usually short: the producer rounds its request up
    // to whole 16384-byte transfers (acquisition.cxx:288-300) and the file ends on a
    // frame.  Those bytes are data, not dropped samples.
    const size_t got = std::fread(buffer.data(), 1, buffer.size(), file_);
    if (got < buffer.size()) exhausted_ = true;
    const size_t usable = got & ~static_cast<size_t>(1);
    if (usable == 0) return false;
    buffer.resize(usable);
    return true;
}

std::unique_ptr<SampleSource> FileSource::clone() const
{
    if (!owns_) return nullptr;                  // stdin cannot be read twice
    std::unique_ptr<SampleSource> copy(new FileSource(path_));
    copy->set_sample_rate(rate_);
    return copy;
}

bool FileSource::position(uint64_t hop_base, uint64_t offset)
{
    if (!owns_) return false;
    exhausted_ = false;
    return fseeko(file_, static_cast<off_t>(hop_base + offset), SEEK_SET) == 0;
}

namespace {

inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

inline int floor_div_256(int v) { return v >= 0 ? v / 256 : -((-v + 255) / 256); }

const int kTone8C[8] = {10, 7, 0, -7, -10, -7, 0, 7};
const int kTone8S[8] = {0, 7, 10, 7, 0, -7, -10, -7};
const int kTone16C[16] = {6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2, 0, 2, 4, 6};
const int kTone16S[16] = {0, 2, 4, 6, 6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2};

inline uint8_t clip_u8(int v) { return static_cast<uint8_t>(std::min(255, std::max(0, v))); }

}  // namespace

void SyntheticSource::generate(uint64_t seed, uint64_t first, uint64_t n, uint8_t* out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = first + i;
        const uint64_t w = splitmix64_at(seed, k);
        int si = 0, sq = 0;
        for (int b = 0; b < 4; ++b) {
            si += static_cast<int>((w >> (8 * b)) & 0xff);
            sq += static_cast<int>((w >> (8 * (b + 4))) & 0xff);
        }
        const int ni = floor_div_256((si - 510) * 35), nq = floor_div_256((sq - 510) * 35);
        const int ti = kTone8C[k % 8] + kTone16C[(3 * k) % 16];
        const int tq = kTone8S[k % 8] + kTone16S[(3 * k) % 16];
        out[2 * i] = clip_u8(128 + ni + ti);
        out[2 * i + 1] = clip_u8(128 + nq + tq);
    }
}

void SyntheticSource::set_frequency(int64_t hz)
{
    frequency_ = hz;
    seed_ = seed_for(base_seed_, hz);
    position_ = 0;
}

std::unique_ptr<SampleSource> SyntheticSource::clone() const
{
    std::unique_ptr<SampleSource> copy(new SyntheticSource(base_seed_));
    copy->set_sample_rate(rate_);
    return copy;
}

bool SyntheticSource::position(uint64_t, uint64_t offset)
{
    position_ = offset / 2;                      // the hop's stream is a function of its frequency alone
    return true;
}

bool SyntheticSource::read(Buffer& buffer)
{
    const uint64_t n = buffer.size() / 2;
    generate(seed_, position_, n, buffer.data());
    position_ += n;
    return true;
}

// ---- live dongle ----------------------------------------------------------------
struct RtlSdrSource::Api {
    void* lib = nullptr;
    uint32_t (*get_device_count)() = nullptr;
    int (*open)(void**, uint32_t) = nullptr;
    int (*close)(void*) = nullptr;
    int (*get_tuner_gains)(void*, int*) = nullptr;
    uint32_t (*get_sample_rate)(void*) = nullptr;
    uint32_t (*get_center_freq)(void*) = nullptr;
    int (*reset_buffer)(void*) = nullptr;
    int (*read_sync)(void*, void*, int, int*) = nullptr;
    int (*set_tuner_gain_mode)(void*, int) = nullptr;
    int (*set_tuner_gain)(void*, int) = nullptr;
    int (*set_center_freq)(void*, uint32_t) = nullptr;
    int (*set_freq_correction)(void*, int) = nullptr;
    int (*set_sample_rate)(void*, uint32_t) = nullptr;
};

namespace {
template <class Fn>
void resolve(void* lib, const char* name, Fn& fn)
{
    fn = reinterpret_cast<Fn>(dlsym(lib, name));
    if (!fn)
        throw RPFexception(std::string("librtlsdr lacks ") + name + ".", ReturnValue::HardwareError);
}
}  // namespace

RtlSdrSource::RtlSdrSource(int dev_index)
{
    api_ = new Api();
    const char* override_path = std::getenv("RPF_RTLSDR_LIB");
    const char* candidates[] = {override_path, "librtlsdr.so.0", "librtlsdr.so"};
    for (const char* c : candidates)
        if (c && *c && (api_->lib = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break;
    if (!api_->lib) {
        delete api_;
        api_ = nullptr;
        throw RPFexception("No RTL-SDR compatible devices found (librtlsdr could not be loaded; "
                           "use --input <file> or --synthetic <seed> to run without a dongle).",
                           ReturnValue::NoDeviceFound);
    }
    try {
        resolve(api_->lib, "rtlsdr_get_device_count", api_->get_device_count);
        resolve(api_->lib, "rtlsdr_open", api_->open);
        resolve(api_->lib, "rtlsdr_close", api_->close);
        resolve(api_->lib, "rtlsdr_get_tuner_gains", api_->get_tuner_gains);
        resolve(api_->lib, "rtlsdr_get_sample_rate", api_->get_sample_rate);
        resolve(api_->lib, "rtlsdr_get_center_freq", api_->get_center_freq);
        resolve(api_->lib, "rtlsdr_reset_buffer", api_->reset_buffer);
        resolve(api_->lib, "rtlsdr_read_sync", api_->read_sync);
        resolve(api_->lib, "rtlsdr_set_tuner_gain_mode", api_->set_tuner_gain_mode);
        resolve(api_->lib, "rtlsdr_set_tuner_gain", api_->set_tuner_gain);
        resolve(api_->lib, "rtlsdr_set_center_freq", api_->set_center_freq);
        resolve(api_->lib, "rtlsdr_set_freq_correction", api_->set_freq_correction);
        resolve(api_->lib, "rtlsdr_set_sample_rate", api_->set_sample_rate);
        // device.cxx:29-50
        const int count = static_cast<int>(api_->get_device_count());
        if (count == 0)
            throw RPFexception("No RTL-SDR compatible devices found.", ReturnValue::NoDeviceFound);
        if (dev_index >= count)
            throw RPFexception("Invalid RTL device number. Only " + std::to_string(count) + " devices available.",
                               ReturnValue::InvalidDeviceIndex);
        if (api_->open(&dev_, static_cast<uint32_t>(dev_index)) < 0)
            throw RPFexception("Could not open rtl_sdr device " + std::to_string(dev_index),
                               ReturnValue::HardwareError);
    } catch (...) {
        dlclose(api_->lib);
        delete api_;
        api_ = nullptr;
        throw;
    }
}

RtlSdrSource::~RtlSdrSource()
{
    if (!api_) return;
    if (dev_) api_->close(dev_);
    dlclose(api_->lib);
    delete api_;
}

std::vector<int> RtlSdrSource::gains() const          // device.cxx:56-70
{
    const int n = api_->get_tuner_gains(dev_, nullptr);
    if (n <= 0)
        throw RPFexception("RTL device: could not read the number of available gains.", ReturnValue::HardwareError);
    std::vector<int> table(n);
    if (api_->get_tuner_gains(dev_, table.data()) <= 0)
        throw RPFexception("RTL device: could not retrieve gain values.", ReturnValue::HardwareError);
    return table;
}

int RtlSdrSource::nearest_gain(int gain) const        // device.cxx:140-151: first of equally near ones
{
    int best = std::numeric_limits<int>::max(), selected = 0;
    for (int g : gains()) {
        const int d = std::abs(g - gain);
        if (d < best) {
            best = d;
            selected = g;
        }
    }
    return selected;
}

void RtlSdrSource::print_gains() const                // device.cxx:153-163
{
    const std::vector<int> table = gains();
    std::cerr << "Available gains (in 1/10th of dB): ";
    for (size_t i = 0; i < table.size(); ++i) std::cerr << (i ? ", " : "") << table[i];
    std::cerr << std::endl;
}

void RtlSdrSource::set_gain(int gain)                 // device.cxx:99-109: manual mode, then the value
{
    int status = api_->set_tuner_gain_mode(dev_, 1);
    status += api_->set_tuner_gain(dev_, gain);
    if (status != 0) throw RPFexception("RTL device: could not set gain.", ReturnValue::HardwareError);
}

void RtlSdrSource::set_freq_correction(int ppm_error) // device.cxx:125-131
{
    if (api_->set_freq_correction(dev_, ppm_error) < 0)
        throw RPFexception("RTL device: could not set frequency correction.", ReturnValue::HardwareError);
}

void RtlSdrSource::set_sample_rate(uint32_t rate)     // device.cxx:133-139
{
    if (api_->set_sample_rate(dev_, rate))
        throw RPFexception("RTL device: could not set sample rate.", ReturnValue::HardwareError);
}

int RtlSdrSource::sample_rate() const                 // device.cxx:72-80
{
    const uint32_t rate = api_->get_sample_rate(dev_);
    if (rate == 0) throw RPFexception("RTL device: could not read sample rate.", ReturnValue::HardwareError);
    return static_cast<int>(rate);
}

void RtlSdrSource::set_frequency(int64_t hz)          // device.cxx:111-123, settling pause included
{
    if (api_->set_center_freq(dev_, static_cast<uint32_t>(hz)) < 0)
        throw RPFexception("RTL device: could not set center frequency.", ReturnValue::HardwareError);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
}

int64_t RtlSdrSource::frequency() const               // device.cxx:82-90
{
    const uint32_t hz = api_->get_center_freq(dev_);
    if (hz == 0) throw RPFexception("RTL device: could not read frequency.", ReturnValue::HardwareError);
    return hz;
}

bool RtlSdrSource::read(Buffer& buffer)               // device.cxx:92-97
{
    int n_read = 0;
    api_->reset_buffer(dev_);
    api_->read_sync(dev_, buffer.data(), static_cast<int>(buffer.size()), &n_read);
    return n_read == static_cast<int>(buffer.size());
}

}  // namespace rpf_host
