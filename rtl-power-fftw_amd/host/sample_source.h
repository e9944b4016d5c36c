// sample_source.h -- where the IQ bytes come from.  Stands where `class Rtlsdr`
// stands in the reference (/root/reference/src/device.h:28-54): tune, report the
// sample rate, fill a Buffer.  Three implementations: a live dongle through
// librtlsdr resolved at run time (no link-time dependency: there is none next
// to an MI355X), a replayed byte stream (file / stdin) and a synthetic receiver.
#ifndef RPF_HOST_SAMPLE_SOURCE_H
#define RPF_HOST_SAMPLE_SOURCE_H

#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "datastore.h"

namespace rpf_host {

class SampleSource {
public:
    virtual ~SampleSource() {}
    virtual void set_sample_rate(uint32_t rate) { rate_ = rate; }
    virtual int sample_rate() const { return static_cast<int>(rate_); }
    virtual void set_frequency(int64_t hz) { frequency_ = hz; }
    virtual int64_t frequency() const { return frequency_; }
    // Fill the buffer (buffer.size() bytes); false = dropped samples / nothing read.
    // A finite replay that hits its end keeps what it got: it shrinks the buffer
    // to the (even) number of bytes read and returns true while that is non-zero.
    virtual bool read(Buffer& buffer) = 0;
    // After a failed read the reference simply tries again (a dongle keeps
    // streaming, acquisition.cxx:307-316); a finite replay has nothing more to give.
    virtual bool retry_after_short_read() const { return true; }
    // A finite replay has delivered its last byte: no later acquisition can succeed.
    virtual bool exhausted() const { return false; }
    // Multi-device scans: every device reads its own shard through
    // its own source.  clone() = an independent source over the same data (nullptr:
    // the source cannot be shared -- one dongle, stdin); position() = make the next
    // read() return the bytes a single sequential reader would see `offset` bytes
    // into the hop that starts `hop_base` bytes into the replay (call after
    // set_frequency; false = not seekable).
    virtual std::unique_ptr<SampleSource> clone() const { return nullptr; }
    virtual bool position(uint64_t hop_base, uint64_t offset) { (void)hop_base; (void)offset; return false; }
protected:
    uint32_t rate_ = 2000000;
    int64_t frequency_ = 0;
};

// Interleaved u8 I/Q from a file or stdin ("-"); a short read ends the data.
class FileSource : public SampleSource {
public:
    explicit FileSource(const std::string& path);
    ~FileSource() override;
    bool read(Buffer& buffer) override;
    bool retry_after_short_read() const override { return false; }
    bool exhausted() const override { return exhausted_; }
    std::unique_ptr<SampleSource> clone() const override;
    bool position(uint64_t hop_base, uint64_t offset) override;
private:
    std::string path_;
    std::FILE* file_ = nullptr;
    bool owns_ = false;
    bool exhausted_ = false;
};

// The integer-only receiver model of rtl-power-fftw_amd/synth.py (noise_tones_iq):
// identical bytes for the same seed, so the CLI can be checked against Python.
// Tuning restarts the stream with a seed that depends only on the frequency,
// seed_for(base, hz) = base + hz mod 9973, so every hop of a scan sees different
// but reproducible data.
class SyntheticSource : public SampleSource {
public:
    explicit SyntheticSource(uint64_t seed) : base_seed_(seed), seed_(seed) {}
    void set_frequency(int64_t hz) override;
    bool read(Buffer& buffer) override;
    std::unique_ptr<SampleSource> clone() const override;
    bool position(uint64_t hop_base, uint64_t offset) override;
    static uint64_t seed_for(uint64_t base, int64_t hz) { return base + static_cast<uint64_t>(hz % 9973); }
    static void generate(uint64_t seed, uint64_t first_sample, uint64_t nsamples, uint8_t* out);
private:
    uint64_t base_seed_, seed_;
    uint64_t position_ = 0;          // complex samples produced since the last tune
};

// A live RTL-SDR dongle, the reference's only source (src/device.cxx:29-151).
// librtlsdr is dlopen()ed (RPF_RTLSDR_LIB, else librtlsdr.so.0 / librtlsdr.so) and
// its 13 entry points resolved by name; a missing library or no dongle ends the
// program with the reference's own NoDeviceFound exit code.
class RtlSdrSource : public SampleSource {
public:
    explicit RtlSdrSource(int dev_index);
    ~RtlSdrSource() override;
    void set_sample_rate(uint32_t rate) override;
    int sample_rate() const override;
    void set_frequency(int64_t hz) override;
    int64_t frequency() const override;
    bool read(Buffer& buffer) override;
    std::vector<int> gains() const;              // tenths of a dB
    int nearest_gain(int gain) const;
    void print_gains() const;
    void set_gain(int gain);
    void set_freq_correction(int ppm_error);
private:
    struct Api;
    Api* api_ = nullptr;
    void* dev_ = nullptr;
};

}  // namespace rpf_host
#endif
