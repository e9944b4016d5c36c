// datastore.h -- C++11 host-side counterpart of the reference's `class Datastore`
// (/root/reference/src/datastore.h:35-68) over the C-ABI of include/rpf_engine.h.
//
// A maintainer of rtl_power_fftw swaps the reference's datastore.{h,cxx} for this
// header and links librpf_engine.so; Acquisition::run keeps its structure, the
// mutex/deque hand-off of acquisition.cxx:278-285,310-314,320-323,343-347
// becomes acquire()/unget()/submit()/finish() (INTEGRATION.md shows the diff).
// Errors surface as RPFexception with the reference's own ReturnValue codes.
#ifndef RPF_HOST_DATASTORE_H
#define RPF_HOST_DATASTORE_H

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rpf_engine.h"

namespace rpf_host {

// exceptions.h:25-34
enum class ReturnValue {
  Success = 0, NoDeviceFound = 1, InvalidDeviceIndex = 2, InvalidArgument = 3,
  TCLAPerror = 4, InvalidInput = 5, AcquisitionError = 6, HardwareError = 7
};

// exceptions.h:39-47
class RPFexception : public std::runtime_error {
public:
  explicit RPFexception(const std::string& what, ReturnValue retval_)
    : runtime_error(what), retval(retval_) {}
  ReturnValue returnValue() const { return retval; }
private:
  ReturnValue retval;
};

// The Params fields the hot path reads (params.h:33-66), same names and defaults.
struct Params {
  int N = 512;
  int buffers = 5;
  int buf_length = 16384 * 100;
  bool window = false;
  int64_t repeats = buf_length / (2 * N);
  int sample_rate = 2000000;
  int64_t cfreq = 1420405752;
  bool linear = false;
  bool baseline = false;
  int device = 0;          // additive: HIP device ordinal
};

// A filled/empty hand-off buffer: what `Buffer&` is in acquisition.cxx:283,302-304
// (data()/size()/resize()), backed by engine-owned pinned memory.
class Buffer {
public:
  uint8_t* data() { return ptr_; }
  const uint8_t* data() const { return ptr_; }
  size_t size() const { return size_; }
  size_t capacity() const { return capacity_; }
  void resize(size_t n) {
    if (n > capacity_)
      throw RPFexception("Buffer::resize beyond buf_length", ReturnValue::InvalidArgument);
    size_ = n;
  }
private:
  friend class Datastore;
  uint8_t* ptr_ = nullptr;
  size_t size_ = 0, capacity_ = 0;
};

class Datastore {
public:
  const Params& params;
  int64_t repeats_done = 0;          // datastore.h:38
  std::vector<double> pwr;           // datastore.h:53 (valid after finish())

  // datastore.cxx:23-34
  // device_override >= 0: HIP device for this instance (one Datastore per device in a
  // multi-device scan) instead of params.device
  Datastore(const Params& params_, std::vector<float>& window_values, int device_override = -1)
    : params(params_), pwr(params_.N) {
    if (params.window && (int)window_values.size() != params.N)
      throw RPFexception("Error reading window function. Expected " + std::to_string(params.N)
                         + " values, found " + std::to_string(window_values.size()) + ".",
                         ReturnValue::InvalidInput);
    rpf_config cfg;
    cfg.struct_size = sizeof(cfg);
    cfg.N = params.N;
    cfg.window = params.window ? window_values.data() : nullptr;
    cfg.n_buffers = params.buffers;
    cfg.buffer_capacity = params.buf_length;
    cfg.device = device_override >= 0 ? device_override : params.device;
    cfg.flags = RPF_FLAG_NONE;
    int rc = rpf_engine_create(&cfg, &engine_);
    if (rc != RPF_OK) throw RPFexception(rpf_last_global_error(), (ReturnValue)rc);
  }
  // datastore.cxx:36-46
  ~Datastore() { rpf_engine_destroy(engine_); }
  Datastore(const Datastore&) = delete;
  Datastore(Datastore&&) = delete;
  Datastore& operator=(const Datastore&) = delete;
  Datastore& operator=(Datastore&&) = delete;

  // acquisition.cxx:252-256: zero pwr, repeats_done = 0, start the worker
  void begin() { begin(params.repeats); }
  // the same for a shard of an acquisition (multi-device scans: this device's share of the frames)
  void begin(int64_t repeats) { check(rpf_begin(engine_, repeats)); repeats_done = 0; }
  // acquisition.cxx:278-285
  Buffer acquire() {
    Buffer b;
    check(rpf_buffer_acquire(engine_, &b.ptr_, &b.capacity_));
    b.size_ = b.capacity_;
    return b;
  }
  // acquisition.cxx:310-314
  void unget(Buffer& b) { check(rpf_buffer_unget(engine_, b.ptr_)); }
  // acquisition.cxx:320-323
  void submit(Buffer& b) { check(rpf_buffer_submit(engine_, b.ptr_, b.size_)); }
  // acquisition.cxx:343-347; pwr and repeats_done are final afterwards
  void finish() {
    check(rpf_finish(engine_, &repeats_done));
    check(rpf_get_power(engine_, pwr.data()));
    // 65536 ... 262144 bins: a launch of the persistent four-step kernel that could not get the whole device (another
    // process on it) gives up; the engine has run those bytes through its two-kernel path and keeps to it -- the
    // acquisition is complete and right, the operator is told once why the rest of the run is a few per cent slower
    int64_t gave_up = 0;
    if (rpf_fused_status(engine_, nullptr, &gave_up, nullptr) == RPF_OK && gave_up > fused_gave_up_) {
      if (fused_gave_up_ == 0)
        std::cerr << "Note: the device was busy; the FFT worker left its single-launch kernel for the two-kernel path "
                     "(results are unaffected)." << std::endl;
      fused_gave_up_ = gave_up;
    }
  }
  // the engine behind this Datastore (multi-device scans hand it to the scan reducer)
  const rpf_engine* engine() const { return engine_; }
  // datastore.cxx:98-103
  void printQueueHistogram() const {
    std::vector<int> h(params.buffers + 1);
    rpf_get_histogram(engine_, h.data());
    std::cerr << "Buffer queue histogram: ";
    for (auto size : h) std::cerr << size << " ";
    std::cerr << std::endl;
  }

private:
  void check(int rc) const {
    if (rc != RPF_OK) throw RPFexception(rpf_last_error(engine_), (ReturnValue)rc);
  }
  rpf_engine* engine_ = nullptr;
  int64_t fused_gave_up_ = 0;
};

}  // namespace rpf_host
#endif  // RPF_HOST_DATASTORE_H
