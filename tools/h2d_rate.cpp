// h2d_rate.cpp -- what pinned-host -> device copies of the reference's default buffer size (1 638 400 B) can reach on this
// box, by number of copy streams and copies in flight: the ceiling of the engine's streaming path.
//   hipcc -O2 tools/h2d_rate.cpp -o tools/h2d_rate && tools/h2d_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// what pinning a caller's pageable stream costs (rpf_accumulate's direct path): hipHostRegister / hipHostUnregister of
// 82 MB (config C2's stream) and of 8 MB pieces, and the copy rate out of registered memory
static int register_cost()
{
    const size_t bytes = 81920000;
    char* p = static_cast<char*>(aligned_alloc(4096, bytes + 4096));
    for (size_t i = 0; i < bytes; i += 4096) p[i] = 1;
    void* dev;
    CHECK(hipMalloc(&dev, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
        const double reg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        CHECK(hipMemcpy(dev, p, bytes, hipMemcpyHostToDevice));
        const double cp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        CHECK(hipHostUnregister(p));
        const double unreg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("hipHostRegister of %zu B: %.2f ms (%.1f GB/s), copy from it %.2f ms (%.1f GB/s), unregister %.2f ms\n", bytes, reg * 1e3,
               bytes / reg / 1e9, cp * 1e3, bytes / cp / 1e9, unreg * 1e3);
    }
    const size_t piece = 8u << 20;
    auto t0 = std::chrono::steady_clock::now();
    size_t n = 0;
    for (size_t off = 0; off + piece <= bytes; off += piece, ++n) CHECK(hipHostRegister(p + off, piece, hipHostRegisterDefault));
    const double reg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    for (size_t off = 0; off + piece <= bytes; off += piece) CHECK(hipHostUnregister(p + off));
    const double unreg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("in %zu pieces of %zu B: register %.2f ms (%.1f GB/s), unregister %.2f ms\n", n, piece, reg * 1e3, n * piece / reg / 1e9, unreg * 1e3);
    // pageable copy (what hipMemcpyAsync does with unregistered memory: staged by the runtime)
    t0 = std::chrono::steady_clock::now();
    CHECK(hipMemcpy(dev, p, bytes, hipMemcpyHostToDevice));
    const double cp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("hipMemcpy of %zu pageable bytes: %.2f ms (%.1f GB/s)\n", bytes, cp * 1e3, bytes / cp / 1e9);
    CHECK(hipFree(dev));
    free(p);
    return 0;
}

int main()
{
    if (register_cost()) return 1;
    const size_t sizes[] = {1638400, 4 * 1638400, 104857600};
    for (size_t bytes : sizes) {
        const int nbuf = 8;
        std::vector<void*> host(nbuf);
        for (auto& h : host) CHECK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        void* dev;
        CHECK(hipMalloc(&dev, bytes * nbuf));
        for (int nstreams : {1, 2, 4}) {
            std::vector<hipStream_t> st(nstreams);
            for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            const int total = bytes > (50u << 20) ? 24 : 400;
            for (int pass = 0; pass < 2; ++pass) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < total; ++i)
                    CHECK(hipMemcpyAsync(static_cast<char*>(dev) + (i % nbuf) * bytes, host[i % nbuf], bytes, hipMemcpyHostToDevice, st[i % nstreams]));
                for (auto& s : st) CHECK(hipStreamSynchronize(s));
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (pass) printf("%9zu B per copy, %d stream(s), %3d copies queued at once: %6.1f GB/s\n", bytes, nstreams, total, total * bytes / sec / 1e9);
            }
            // the same with what the engine adds per buffer: an event after every copy (every 2nd, every 4th), and a thread
            // that waits for each event in issue order (the recycler)
            if (bytes == 1638400 && nstreams == 2) {
                std::vector<hipEvent_t> ev(400);
                for (auto& evt : ev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
                for (int every : {1, 2, 4}) {
                    for (int pass = 0; pass < 2; ++pass) {
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int i = 0; i < total; ++i) {
                            CHECK(hipMemcpyAsync(static_cast<char*>(dev) + (i % nbuf) * bytes, host[i % nbuf], bytes, hipMemcpyHostToDevice, st[i % nstreams]));
                            if (i % every == every - 1) CHECK(hipEventRecord(ev[i], st[i % nstreams]));
                        }
                        for (int i = every - 1; i < total; i += every) CHECK(hipEventSynchronize(ev[i]));
                        for (auto& s : st) CHECK(hipStreamSynchronize(s));
                        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        if (pass) printf("%9zu B per copy, %d streams, an event recorded after every %d copies and waited for: %6.1f GB/s\n", bytes, nstreams, every, total * bytes / sec / 1e9);
                    }
                }
                // bounded pipeline: at most `depth` copies in flight (what n buffers in the pool allow), the next one issued
                // when the oldest has landed; stream = the one with fewer copies in flight / alternating
                for (int depth : {3, 4, 5, 8}) {
                    for (int policy = 0; policy < 2; ++policy) {
                        int inflight[2] = {0, 0};
                        std::vector<int> on(total);
                        int head = 0;
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int i = 0; i < total; ++i) {
                            if (i - head >= depth) {
                                CHECK(hipEventSynchronize(ev[head]));
                                --inflight[on[head]];
                                ++head;
                            }
                            const int sidx = policy ? (inflight[0] <= inflight[1] ? 0 : 1) : i % 2;
                            on[i] = sidx;
                            ++inflight[sidx];
                            CHECK(hipMemcpyAsync(static_cast<char*>(dev) + (i % nbuf) * bytes, host[i % nbuf], bytes, hipMemcpyHostToDevice, st[sidx]));
                            CHECK(hipEventRecord(ev[i], st[sidx]));
                        }
                        for (auto& s : st) CHECK(hipStreamSynchronize(s));
                        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        printf("%9zu B per copy, 2 streams, at most %d copies in flight, %s: %6.1f GB/s\n", bytes, depth,
                               policy ? "least-loaded stream" : "alternating streams", total * bytes / sec / 1e9);
                    }
                }
                for (auto& evt : ev) CHECK(hipEventDestroy(evt));
            }
            for (auto& s : st) CHECK(hipStreamDestroy(s));
        }
        CHECK(hipFree(dev));
        for (auto& h : host) CHECK(hipHostFree(h));
    }
    return 0;
}
