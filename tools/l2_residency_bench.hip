// l2_residency_bench.hip -- does data one CU stores stay in its XCD's L2 for another CU of the SAME XCD
// to read, and do the stored bytes leave the L2 anyway?  (the question behind a fused
// four-step kernel whose intermediate Y never crosses the fabric)
//
//   hipcc --offload-arch=gfx950 -O3 tools/l2_residency_bench.hip -o tools/l2_residency_bench
//   tools/l2_residency_bench                      # times and stale-read counts, every variant
//   tools/gpu_l2_residency.sh                     # the same under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
//
// One persistent launch, one 256-thread workgroup per CU (96 KB of LDS requested so that no two share a CU).
// The workgroups that find themselves on one XCD (HW_REG_XCC_ID) form a team: ranks 0-7 write, 8-15 read,
// 16-31 idle (the ALL-32 rows below use every workgroup).  Per round: writers store the team's buffer (16 B per lane, pattern = f(round, index)),
// drain (s_waitcnt vmcnt(0)), arrive on the team's `produced` counter; readers wait for all writers, apply the
// acquire under test, read the whole buffer with 8 x 16 B loads in flight per lane, count words that do not
// carry this round's pattern (= stale reads), arrive on `consumed`; writers wait for that before the next round.
// The counters are touched only by L2-executed atomics.  Every spin is bounded.
//
// Each (store flavour, load flavour, -, cross) combination is its own kernel NAME so that rocprofv3's
// per-kernel counters can be read apart.  CROSS = readers read the buffer of XCD (x+1)%8: the calibration row
// (those reads cannot be L2 hits; FETCH_SIZE must show footprint x rounds, in whatever unit it really counts).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kWriters = 8, kReaders = 8;
constexpr unsigned kSpinLimit = 1u << 22;
#ifndef POLL_ATOMICS
#define POLL_ATOMICS 0
#endif
constexpr bool kPollWithAtomics = POLL_ATOMICS;   // -DPOLL_ATOMICS=1: poll with returning atomic adds of zero (hot-spots one L2 channel)
#ifndef POLL_SLEEP
#define POLL_SLEEP 8
#endif

struct Ctl {
    unsigned members[8][32];     // [xcd][0]: workgroups registered on that XCD
    unsigned produced[8][32];    // [xcd][0]
    unsigned consumed[8][32];    // [xcd][0]
    unsigned registered[32];
    unsigned abort_[32];
    unsigned long long stale[256];   // per workgroup (team rank + 32 x xcd): words read that were not this round's
    unsigned long long ticks[256];   // per workgroup: 100 MHz ticks spent storing + draining (writers) / loading (readers)
};

enum StoreKind { ST_PLAIN, ST_NT, ST_SC1, ST_SC0SC1 };
enum LoadKind { LD_INV_PLAIN, LD_SC1, LD_PLAIN_NOINV, LD_SC0SC1 };

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void store16(u4* p, u4 v)
{
    if constexpr (K == ST_PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (K == ST_NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (K == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// Eight 16-byte loads in flight and the wait for them in ONE asm statement: the compiler does not know that an
// asm load's destination is written later, when the data returns -- given the chance it copies or reuses those
// registers before a separate s_waitcnt (seen here as a memory aperture violation: an address register overwritten
// by a returning load).
#define LOAD8(FLAVOUR)                                                                                                   \
    asm volatile("global_load_dwordx4 %0, %8, off" FLAVOUR "\n\tglobal_load_dwordx4 %1, %9, off" FLAVOUR                  \
                 "\n\tglobal_load_dwordx4 %2, %10, off" FLAVOUR "\n\tglobal_load_dwordx4 %3, %11, off" FLAVOUR           \
                 "\n\tglobal_load_dwordx4 %4, %12, off" FLAVOUR "\n\tglobal_load_dwordx4 %5, %13, off" FLAVOUR           \
                 "\n\tglobal_load_dwordx4 %6, %14, off" FLAVOUR "\n\tglobal_load_dwordx4 %7, %15, off" FLAVOUR           \
                 "\n\ts_waitcnt vmcnt(0)"                                                                                \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])                \
                 : "memory")
template <int K>
__device__ __forceinline__ void load8(u4 (&v)[8], const u4* const (&p)[8])
{
    if constexpr (K == LD_SC1) LOAD8(" sc1");
    else if constexpr (K == LD_SC0SC1) LOAD8(" sc0 sc1");
    else LOAD8("");
}

// AGENT: the counter is shared with another XCD (CROSS rows): the atomic must execute at the memory-side
// coherence point (sc1), not in this XCD's L2.
template <bool AGENT>
__device__ __forceinline__ unsigned l2_atomic_read(unsigned* p)
{
    unsigned r;
    const unsigned zero = 0;
    if (kPollWithAtomics) {
        if constexpr (AGENT) asm volatile("global_atomic_add %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(zero) : "memory");
        else asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(zero) : "memory");
    } else {
        // an sc1 load is served by the L2 (never by this CU's L1), where the team's atomics execute; for a counter
        // another XCD updates (AGENT) sc0 sc1 goes to the memory-side coherence point
        if constexpr (AGENT) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    }
    return r;
}
template <bool AGENT>
__device__ __forceinline__ bool wait_for(unsigned* ctr, unsigned target, Ctl* ctl, int* flag)
{
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int ok = 1;
        while (l2_atomic_read<AGENT>(ctr) < target) {
            __builtin_amdgcn_s_sleep(POLL_SLEEP);
            if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(&ctl->abort_[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}
template <bool AGENT>
__device__ __forceinline__ void arrive(unsigned* ctr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if constexpr (AGENT) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__device__ __forceinline__ u4 pattern(unsigned round, unsigned idx)
{
    const unsigned h = round * 0x9E3779B1u + idx * 0x85EBCA77u;
    return u4{h, h ^ 0xdeadbeefu, round, idx};
}

// bytes: the team's buffer (a multiple of 4096); buf: 8 such buffers 16 MB apart; table, sink: unused.
// (FOREIGN: a third group of workgroups streaming a 2 MB table through the same L2 was tried and removed -- its sweep
//  faulted on this stack, and the round-4 fused kernel has no such stream; the parameter stays for the kernel names in
//  profiles/r04_l2_residency_raw.txt)
template <int ST, int LD, bool FOREIGN, bool CROSS>
__global__ __launch_bounds__(256) void residency_kernel(u4* __restrict__ buf, size_t bytes, int rounds,
                                                        const u4* __restrict__ table, Ctl* __restrict__ ctl, unsigned* sink, int stage)
{
    extern __shared__ unsigned char smem[];
    __shared__ int team[4];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcd = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20)) & 7;
        const unsigned rank = __hip_atomic_fetch_add(&ctl->members[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->registered[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(&ctl->registered[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { ok = 0; break; }
        }
        for (int x = 0; ok && x < 8; ++x)
            if (__hip_atomic_load(&ctl->members[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32u) ok = 0;
        if (!ok) __hip_atomic_store(&ctl->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        team[0] = xcd;
        team[1] = static_cast<int>(rank);
        team[2] = ok;
        smem[0] = 0;                                                // the LDS request is real
    }
    __syncthreads();
    const int xcd = team[0], rank = team[1];
    if (!team[2] || stage == 0) return;
    constexpr size_t kTeamStride = (16u << 20) / 16;               // in 16-byte units
    u4* const mine = buf + kTeamStride * xcd;
    const u4* const theirs = buf + kTeamStride * (CROSS ? (xcd + 1) % 8 : xcd);
    unsigned* const produced = &ctl->produced[xcd][0];
    unsigned* const consumed = &ctl->consumed[xcd][0];
    // CROSS: the buffer a reader reads is written by the NEXT XCD's writers: wait on their counter, and
    // report consumption to them
    unsigned* const their_produced = &ctl->produced[CROSS ? (xcd + 1) % 8 : xcd][0];
    unsigned* const their_consumed = &ctl->consumed[CROSS ? (xcd + 1) % 8 : xcd][0];
    const int nchunks = static_cast<int>(bytes / 4096);            // 256 lanes x 16 B

    if (rank < kWriters) {
        unsigned long long cyc = 0;
        for (int r = 0; r < rounds; ++r) {
            if (stage >= 2 && r > 0 && !wait_for<CROSS>(consumed, static_cast<unsigned>(kReaders) * r, ctl, &team[3])) return;
            const unsigned long long t0 = wall_clock64();
            for (int c = rank; c < nchunks; c += kWriters) {
                const unsigned idx = static_cast<unsigned>(c) * 256u + tid;
                store16<ST>(mine + idx, pattern(static_cast<unsigned>(r), idx));
            }
            arrive<CROSS>(produced);
            cyc += wall_clock64() - t0;
        }
        if (tid == 0) ctl->ticks[32 * xcd + rank] = cyc;
    } else if (rank < kWriters + kReaders && stage >= 2) {
        const int rd = rank - kWriters;
        unsigned long long bad = 0, cyc = 0;
        for (int r = 0; r < rounds; ++r) {
            if (!wait_for<CROSS>(their_produced, static_cast<unsigned>(kWriters) * (r + 1), ctl, &team[3])) return;
            const unsigned long long t0 = wall_clock64();
            if constexpr (LD == LD_INV_PLAIN) asm volatile("buffer_inv sc1" ::: "memory");
            for (int c0 = rd; c0 < nchunks; c0 += kReaders * 8) {          // (nchunks is a multiple of kReaders * 8)
                u4 v[8];
                const u4* p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = theirs + static_cast<unsigned>(c0 + kReaders * u) * 256u + tid;
                load8<LD>(v, p);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned idx = static_cast<unsigned>(c0 + kReaders * u) * 256u + tid;
                    const u4 want = pattern(static_cast<unsigned>(r), idx);
                    bad += (v[u].x != want.x) + (v[u].y != want.y) + (v[u].z != want.z) + (v[u].w != want.w);
                }
            }
            cyc += wall_clock64() - t0;
            arrive<CROSS>(their_consumed);
        }
        // (per-thread counts meet in LDS; one word per workgroup leaves the kernel)
        __shared__ unsigned long long wg_bad;
        if (tid == 0) wg_bad = 0;
        __syncthreads();
        if (bad) atomicAdd(&wg_bad, bad);
        __syncthreads();
        if (tid == 0) {
            ctl->stale[32 * xcd + rank] = wg_bad;
            ctl->ticks[32 * xcd + rank] = cyc;
        }
    }
}


// All 32 workgroups of a team write their share of the buffer, then read a NEIGHBOUR's share (what the fused four-step
// kernel's producers and consumers do with Y at full width): how long do the store drain and the read-back of
// bytes / 32 per CU take when every CU of the XCD is at it?  The same kernel signature as above (table, sink unused).
template <int ST, int LD>
__global__ __launch_bounds__(256) void team32_kernel(u4* __restrict__ buf, size_t bytes, int rounds, const u4* __restrict__,
                                                     Ctl* __restrict__ ctl, unsigned*, int stage)
{
    extern __shared__ unsigned char smem[];
    __shared__ int team[4];
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int xcd = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20)) & 7;
        const unsigned rank = __hip_atomic_fetch_add(&ctl->members[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->registered[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(&ctl->registered[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) { ok = 0; break; }
        }
        for (int x = 0; ok && x < 8; ++x)
            if (__hip_atomic_load(&ctl->members[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32u) ok = 0;
        if (!ok) __hip_atomic_store(&ctl->abort_[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        team[0] = xcd;
        team[1] = static_cast<int>(rank);
        team[2] = ok;
        smem[0] = 0;
    }
    __syncthreads();
    const int xcd = team[0], rank = team[1];
    if (!team[2] || stage == 0) return;
    u4* const mine = buf + ((16u << 20) / 16) * static_cast<size_t>(xcd);
    unsigned* const produced = &ctl->produced[xcd][0];
    unsigned* const consumed = &ctl->consumed[xcd][0];
    const int nchunks = static_cast<int>(bytes / 4096);            // a multiple of 32 x 8
    const int src = (rank + 1) & 31;                               // whose chunks this workgroup reads
    unsigned long long bad = 0, wr = 0, rd = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r > 0 && !wait_for<false>(consumed, 32u * r, ctl, &team[3])) return;
        unsigned long long t0 = wall_clock64();
        for (int c = rank; c < nchunks; c += 32) {
            const unsigned idx = static_cast<unsigned>(c) * 256u + tid;
            store16<ST>(mine + idx, pattern(static_cast<unsigned>(r), idx));
        }
        arrive<false>(produced);
        wr += wall_clock64() - t0;
        if (!wait_for<false>(produced, 32u * (r + 1), ctl, &team[3])) return;
        t0 = wall_clock64();
        if constexpr (LD == LD_INV_PLAIN) asm volatile("buffer_inv sc1" ::: "memory");
        for (int c0 = src; c0 < nchunks; c0 += 32 * 8) {
            u4 v[8];
            const u4* p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = mine + static_cast<unsigned>(c0 + 32 * u) * 256u + tid;
            load8<LD>(v, p);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned idx = static_cast<unsigned>(c0 + 32 * u) * 256u + tid;
                const u4 want = pattern(static_cast<unsigned>(r), idx);
                bad += (v[u].x != want.x) + (v[u].y != want.y) + (v[u].z != want.z) + (v[u].w != want.w);
            }
        }
        rd += wall_clock64() - t0;
        arrive<false>(consumed);
    }
    __shared__ unsigned long long wg_bad;
    if (tid == 0) wg_bad = 0;
    __syncthreads();
    if (bad) atomicAdd(&wg_bad, bad);
    __syncthreads();
    if (tid == 0) {
        ctl->stale[32 * xcd + rank] = wg_bad;
        ctl->ticks[32 * xcd + rank] = (wr << 32) | (rd & 0xffffffffull);   // both fit 32 bits (100 MHz ticks)
    }
}

struct Variant {
    bool team32;       // all 32 workgroups of a team write, then read a neighbour's share
    const char* name;
    void (*fn)(u4*, size_t, int, const u4*, Ctl*, unsigned*, int);
};

#define V(st, ld, fo, cr, label) {false, label, residency_kernel<st, ld, fo, cr>}
#define V32(st, ld, label) {true, label, team32_kernel<st, ld>}
static const Variant kVariants[] = {
    V(ST_PLAIN, LD_INV_PLAIN, false, false, "plain stores, buffer_inv sc1 + plain loads"),
    V(ST_PLAIN, LD_SC1, false, false, "plain stores, sc1 loads"),
    V(ST_PLAIN, LD_PLAIN_NOINV, false, false, "plain stores, plain loads, NO acquire (expect stale)"),
    V(ST_NT, LD_INV_PLAIN, false, false, "nt stores, buffer_inv sc1 + plain loads"),
    V(ST_SC1, LD_SC1, false, false, "sc1 stores, sc1 loads"),
    V(ST_SC0SC1, LD_SC0SC1, false, false, "sc0 sc1 stores, sc0 sc1 loads"),
    V(ST_SC1, LD_SC1, false, true, "CROSS-XCD calibration: sc1 stores, sc1 loads of the NEXT XCD's buffer"),
    V32(ST_PLAIN, LD_SC1, "ALL 32 CUs write, then read a neighbour's share: plain stores, sc1 loads"),
    V32(ST_PLAIN, LD_INV_PLAIN, "ALL 32 CUs write, then read a neighbour's share: plain stores, buffer_inv sc1 + plain loads"),
    V32(ST_SC1, LD_SC1, "ALL 32 CUs write, then read a neighbour's share: sc1 stores, sc1 loads (write-through)"),
};

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const int stage = argc > 3 ? atoi(argv[3]) : 2;             // debugging: 0 = team assembly only, 1 = + writers, 2 = everything              // run one variant only (index into kVariants)
    setvbuf(stdout, nullptr, _IONBF, 0);                         // a GPU fault aborts the process: lose nothing
    int dev_cus = 0;
    CHECK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (dev_cus != 256) { printf("needs a 256-CU part (found %d CUs)\n", dev_cus); return 2; }
    u4* buf;
    u4* table;
    Ctl* ctl;
    unsigned* sink;
    CHECK(hipMalloc(&buf, static_cast<size_t>(8) * (16u << 20)));
    CHECK(hipMalloc(&table, static_cast<size_t>(8) * (2u << 20)));
    CHECK(hipMalloc(&ctl, sizeof(Ctl)));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 0x11, static_cast<size_t>(8) * (2u << 20)));
    const size_t lds = 96 * 1024;
    for (const Variant& v : kVariants)
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    printf("# rounds %d; per XCD team: %d writer CUs, %d reader CUs (other CUs of the same XCD)\n", rounds, kWriters, kReaders);
    printf("# footprint = bytes per team x 8 teams; 'stale' = 32-bit words read that were not this round's\n");
    const size_t sizes[] = {1u << 20, 2u << 20, 3u << 20};
    for (size_t bytes : sizes) {
        for (const Variant& v : kVariants) {
            if (only >= 0 && &v != &kVariants[only]) continue;
            printf("%zu KB/team  %-78s ", bytes >> 10, v.name);
            Ctl h;
            float best = 1e30f;
            unsigned long long stale = 0;
            bool aborted = false;
            for (int rep = 0; rep < 2; ++rep) {          // second launch timed; both visible to rocprofv3
                CHECK(hipMemset(buf, 0xff, static_cast<size_t>(8) * (16u << 20)));
                CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(v.fn, dim3(256), dim3(256), lds, 0, buf, bytes, rounds, table, ctl, sink, stage);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, a, b));
                CHECK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
                if (h.abort_[0]) aborted = true;
                for (int w = 0; w < 256; ++w) stale += h.stale[w];
                if (ms < best) best = ms;
            }
            const double us_round = best * 1e3 / rounds;
            double rd_us = 0, wr_us = 0;                                  // mean over the teams' readers / writers
            if (v.team32) {
                for (int w = 0; w < 256; ++w) { wr_us += static_cast<double>(h.ticks[w] >> 32); rd_us += static_cast<double>(h.ticks[w] & 0xffffffffull); }
                rd_us = rd_us / 256.0 / rounds / 100.0;
                wr_us = wr_us / 256.0 / rounds / 100.0;
            } else {
                for (int w = 0; w < 256; ++w) ((w % 32) < kWriters ? wr_us : rd_us) += static_cast<double>(h.ticks[w]);
                rd_us = rd_us / (8.0 * kReaders) / rounds / 100.0;            // 100 MHz counter
                wr_us = wr_us / (8.0 * kWriters) / rounds / 100.0;
            }
            printf("%s %8.3f us/round  (write+drain %6.2f us, read %6.2f us = %6.0f GB/s per XCD)  stale %llu\n",
                   aborted ? "ABORTED" : "ok", us_round, wr_us, rd_us, bytes / (rd_us * 1e-6) / 1e9, stale);
        }
    }
    return 0;
}
