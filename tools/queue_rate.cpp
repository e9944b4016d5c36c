// PCIe-inclusive throughput of the engine's buffer queue driven from C++ (no
// Python in the producer loop): acquire/submit pinned buffers that were filled
// once, like a file replay from page cache.  For DESIGN.md; never bench.py's value.
//   g++ -O2 -std=c++11 tools/queue_rate.cpp -Iinclude -Lrtl-power-fftw_amd -lrpf_engine \
//       -Wl,-rpath,$PWD/rtl-power-fftw_amd -o tools/queue_rate
#include <chrono>
#include <cstdio>
#include <cstring>

#include "rpf_engine.h"

int main()
{
    const int N = 4096;
    const struct { long buf; int nbuf; long repeats; } cases[] = {
        {1638400, 5, 400000}, {1638400, 16, 400000}, {16 * 1638400L, 5, 800000}, {64 * 1638400L, 4, 800000}};
    for (const auto& c : cases) {
        rpf_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.N = N;
        cfg.n_buffers = c.nbuf;
        cfg.buffer_capacity = c.buf;
        rpf_engine* e = nullptr;
        if (rpf_engine_create(&cfg, &e) != RPF_OK) { printf("create: %s\n", rpf_last_global_error()); return 1; }
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            rpf_begin(e, c.repeats);
            const long need = 2L * N * c.repeats;
            long sent = 0;
            int touched = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (sent < need) {
                uint8_t* p; size_t cap;
                if (rpf_buffer_acquire(e, &p, &cap) != RPF_OK) return 2;
                const long n = need - sent < (long)cap ? need - sent : (long)cap;
                if (rep == 0 && touched < c.nbuf) { memset(p, 0x80 + touched, cap); ++touched; }
                if (rpf_buffer_submit(e, p, n) != RPF_OK) return 3;
                sent += n;
            }
            int64_t done = 0;
            rpf_finish(e, &done);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double rate = N * (double)done / dt / 1e9;
            if (rate > best) best = rate;
        }
        printf("buffer %9ld B x %2d: %.1f Gsample/s = %.1f GB/s over PCIe\n", c.buf, c.nbuf, best, 2 * best);
        rpf_engine_destroy(e);
    }
    return 0;
}
