// tsan_queue_driver.cpp -- the engine's producer/consumer hand-off under ThreadSanitizer (VERDICT r03
// item 6).  Protocol being exercised: /root/reference/src/datastore.h:40-47 as flattened in include/rpf_engine.h --
// one producer thread (this one) against the engine's consumer thread, the staging ring and the copy events.
//   make -C rtl-power-fftw_amd/csrc tsan && tools/gpu_tsan.sh          (builds and runs this on the GPU box)
// Scenarios: (1) small buffers that cut frames, several acquisitions on one engine; (2) unget + early finish;
// (3) a zero-length acquisition; (4) rpf_accumulate with 105 MB buffers (fill_buffer's helper threads);
// (5) two engines on one device driven from two threads (MultiDeviceScan's shape).  Every result is compared with
// a second run of the same input: a race that corrupts data shows up even where the sanitizer cannot see it.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rpf_engine.h"

#include <cmath>

static int failures = 0;
// the same stream twice: equal up to the grouping of the f64 partial sums (which launch a frame lands in)
static bool same(const std::vector<double>& a, const std::vector<double>& b)
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (std::fabs(a[i] - b[i]) > 1e-12 * std::fabs(a[i])) return false;
    return true;
}
#define EXPECT(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); ++failures; } } while (0)

static std::vector<uint8_t> make_stream(size_t bytes, uint64_t seed)
{
    std::vector<uint8_t> s(bytes);
    uint64_t z = seed;
    for (size_t i = 0; i < bytes; ++i) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        s[i] = static_cast<uint8_t>(z >> 56);
    }
    return s;
}

static rpf_engine* create(int N, int n_buffers, int64_t capacity)
{
    rpf_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.N = N;
    cfg.n_buffers = n_buffers;
    cfg.buffer_capacity = capacity;
    rpf_engine* e = nullptr;
    const int rc = rpf_engine_create(&cfg, &e);
    if (rc != RPF_OK) printf("rpf_engine_create(N=%d): %d %s\n", N, rc, rpf_last_global_error());
    return rc == RPF_OK ? e : nullptr;
}

// the whole stream through acquire/submit in `piece`-byte readouts
static int64_t feed(rpf_engine* e, const std::vector<uint8_t>& s, size_t piece, int64_t repeats, std::vector<double>* pwr, int N)
{
    EXPECT(rpf_begin(e, repeats) == RPF_OK);
    size_t pos = 0;
    while (pos < s.size()) {
        uint8_t* buf = nullptr;
        size_t cap = 0;
        EXPECT(rpf_buffer_acquire(e, &buf, &cap) == RPF_OK);
        const size_t n = std::min(std::min(piece, cap), (s.size() - pos) & ~static_cast<size_t>(1));
        if (n == 0) { rpf_buffer_unget(e, buf); break; }
        std::memcpy(buf, s.data() + pos, n);
        EXPECT(rpf_buffer_submit(e, buf, n) == RPF_OK);
        pos += n;
    }
    int64_t done = -1;
    EXPECT(rpf_finish(e, &done) == RPF_OK);
    pwr->resize(N);
    EXPECT(rpf_get_power(e, pwr->data()) == RPF_OK);
    return done;
}

int main()
{
    {   // (1) frames straddle 16382-byte buffers; three acquisitions on one engine
        const int N = 4096;
        rpf_engine* e = create(N, 5, 16382);
        if (!e) return 2;
        const std::vector<uint8_t> s = make_stream(static_cast<size_t>(2) * N * 75 + 334, 1);
        std::vector<double> first, again;
        EXPECT(feed(e, s, 16382, 73, &first, N) == 73);
        for (int k = 0; k < 2; ++k) {
            EXPECT(feed(e, s, 16382, 73, &again, N) == 73);
            EXPECT(same(first, again));
        }
        // (2) unget, then an early finish
        EXPECT(rpf_begin(e, 1000) == RPF_OK);
        uint8_t *b = nullptr, *b2 = nullptr;
        EXPECT(rpf_buffer_acquire(e, &b, nullptr) == RPF_OK);
        EXPECT(rpf_buffer_unget(e, b) == RPF_OK);
        EXPECT(rpf_buffer_acquire(e, &b2, nullptr) == RPF_OK && b2 == b);
        std::memcpy(b2, s.data(), 16382);
        EXPECT(rpf_buffer_submit(e, b2, 16382) == RPF_OK);
        int64_t done = -1;
        EXPECT(rpf_finish(e, &done) == RPF_OK && done == 1);
        // (3) zero-length acquisition
        EXPECT(rpf_begin(e, 0) == RPF_OK);
        EXPECT(rpf_finish(e, &done) == RPF_OK && done == 0);
        int hist[6];
        EXPECT(rpf_get_histogram(e, hist) == RPF_OK);
        rpf_engine_destroy(e);
    }
    {   // (4) rpf_accumulate, 5 x 105 MB buffers: fill_buffer forks helper threads per buffer
        const int N = 4096;
        const int64_t cap = 105 << 20;
        rpf_engine* e = create(N, 5, cap);
        if (!e) return 2;
        const std::vector<uint8_t> s = make_stream(static_cast<size_t>(3) * cap + 5000, 2);
        const int64_t frames = static_cast<int64_t>(s.size() / (2 * N));
        std::vector<double> a(N), b(N);
        int64_t da = 0, db = 0;
        EXPECT(rpf_accumulate(e, s.data(), s.size(), frames, a.data(), &da) == RPF_OK && da == frames);
        EXPECT(rpf_accumulate(e, s.data(), s.size(), frames, b.data(), &db) == RPF_OK && db == frames);
        EXPECT(same(a, b));
        rpf_engine_destroy(e);
    }
    {   // (5) two engines, two producer threads
        const int N = 1024;
        rpf_engine* e0 = create(N, 3, 65536);
        rpf_engine* e1 = create(N, 3, 65536);
        if (!e0 || !e1) return 2;
        const std::vector<uint8_t> s = make_stream(static_cast<size_t>(2) * N * 400, 3);
        std::vector<double> p0, p1;
        int64_t d0 = 0, d1 = 0;
        std::thread t0([&]() { for (int k = 0; k < 3; ++k) d0 = feed(e0, s, 65536, 400, &p0, N); });
        std::thread t1([&]() { for (int k = 0; k < 3; ++k) d1 = feed(e1, s, 65536, 400, &p1, N); });
        t0.join();
        t1.join();
        EXPECT(d0 == 400 && d1 == 400 && same(p0, p1));
        rpf_engine_destroy(e0);
        rpf_engine_destroy(e1);
    }
    printf("tsan_queue_driver: %s\n", failures ? "FAILURES" : "all scenarios ok");
    return failures ? 1 : 0;
}
