// h2d_rate.cpp -- what pinned-host -> device copies of the reference's default buffer size (1 638 400 B) can reach on this
// box, by number of copy streams and copies in flight: the ceiling of the engine's streaming path.
//   hipcc -O2 tools/h2d_rate.cpp -o tools/h2d_rate && tools/h2d_rate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
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
            for (auto& s : st) CHECK(hipStreamDestroy(s));
        }
        CHECK(hipFree(dev));
        for (auto& h : host) CHECK(hipHostFree(h));
    }
    return 0;
}
