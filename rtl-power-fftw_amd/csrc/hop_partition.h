// hop_partition.h -- how ONE persistent launch of the fused kernel walks several
// acquisitions ("hops" of a scan: /root/reference/src/rtl_power_fftw.cxx:133-174 loops
// over them, acquisition.cxx:252-256 zeroes pwr and restarts the worker for each).
//
// Unit of work = one "iteration": the FPW frames a workgroup transforms side by side,
// all from the same hop (hop h has ceil(nframes[h] / FPW) iterations).  The iterations
// of all hops of the launch form one sequence of I = q G + r iterations; workgroup w of G
// runs q (+1 if w < r) of them and hands over one partial spectrum per hop it touched
// ("segment"):
//   * scans (H > 1, step = 1): the contiguous range that starts at w q + min(w, r), so a
//     workgroup meets a hop boundary at most every q iterations.  Segments are numbered in
//     sequence order: the partial spectra of hop h are the contiguous slots
//     [slot_begin[h], slot_begin[h+1]) and the reduce kernel adds them in a fixed order;
//   * (H = 1, step = G, optional: iterations w, w + G, w + 2G, ... -- the order of the
//     single-acquisition kernel, kept for A/B measurements of the two orders; slot = w.)
//
// Plain C++: the host side of the engine, the kernel (through a lane-resident copy of the
// tables, rpf_kernels.hip) and the CPU tests (tests/emul) share the cursor below.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define RPF_HOP_HD __host__ __device__ __forceinline__
#else
#define RPF_HOP_HD inline
#endif

namespace rpf {

constexpr int kMaxHops = 16;     // hops per launch (kernel arguments by value); longer scans take several launches

struct HopArgs {
    int H;                               // hops in this launch, 1 .. kMaxHops
    int total;                           // iterations in the launch = it_begin[H] = q grid + r
    int q, r;                            // (no division on the device)
    int step;                            // distance between a workgroup's iterations: 1 (or the grid, see above)
    int pad_;
    int nframes[kMaxHops];               // frames of hop h
    int it_begin[kMaxHops + 1];          // first iteration of hop h; entries H .. kMaxHops = total
    int slot_bias[kMaxHops];             // partial slot of (workgroup w, hop h) = slot_bias[h] + w
    const uint8_t* stream[kMaxHops];     // first byte of hop h's frames (frame f = bytes [2N f, 2N (f + 1)))
};

struct SlotRanges {
    int begin[kMaxHops + 1];             // partial slots of hop h = [begin[h], begin[h+1])
};

// iterations [lo, hi) of workgroup w in the contiguous form
RPF_HOP_HD void hop_range(int w, int q, int r, int* lo, int* hi)
{
    *lo = w * q + (w < r ? w : r);
    *hi = *lo + q + (w < r ? 1 : 0);
}
// first iteration and number of iterations of workgroup w (its k-th iteration = first + k step)
RPF_HOP_HD void hop_share(int w, int q, int r, int step, int* first, int* count)
{
    *first = step == 1 ? w * q + (w < r ? w : r) : w;
    *count = q + (w < r ? 1 : 0);
}

// Where a workgroup stands in the launch's iteration sequence: iteration j of hop h.  TABLE
// answers it_begin(h), nframes(h), stream(h) and hop_of(iteration): HopArgsView below on the
// host, a lane-resident copy read with v_readlane in the kernel -- either way the cursor's
// fields are wave-uniform and are refreshed only when the cursor enters another hop.
struct HopCursor {
    int j, h, begin, end, nframes;
    const uint8_t* stream;
    template <class TABLE>
    RPF_HOP_HD void seek(const TABLE& tbl, int it)
    {
        j = it;
        h = tbl.hop_of(it);
        begin = tbl.it_begin(h);
        end = tbl.it_begin(h + 1);
        nframes = tbl.nframes(h);
        stream = tbl.stream(h);
    }
};

struct HopArgsView {
    const HopArgs& a;
    int it_begin(int h) const { return a.it_begin[h]; }
    int nframes(int h) const { return a.nframes[h]; }
    const uint8_t* stream(int h) const { return a.stream[h]; }
    // the hop that holds iteration `it` (< total) = the number of hop starts 1 .. kMaxHops at or
    // before it (empty hops share their successor's start and are skipped; entries past H = total)
    int hop_of(int it) const
    {
        int hop = 0;
        for (int h = 1; h <= kMaxHops; ++h) hop += a.it_begin[h] <= it ? 1 : 0;
        return hop;
    }
};

// Fills *a (all but a->stream, which is the caller's) and the slot ranges for `H` hops of
// `nframes[h]` frames on a grid of at most `max_grid` workgroups running `fpw` frames each.
// Returns the grid to launch (0: no frame at all -- nothing to launch, every slot range is
// empty) or -1 if the arguments do not fit (H, or 2^31 iterations).
inline int partition_hops(const int64_t* nframes, int H, int fpw, int max_grid, HopArgs* a, SlotRanges* r,
                          bool interleave_single = false)
{
    if (H < 1 || H > kMaxHops || fpw < 1 || max_grid < 1) return -1;
    a->H = H;
    int64_t it = 0;
    for (int h = 0; h < H; ++h) {
        if (nframes[h] < 0 || nframes[h] > INT32_MAX) return -1;
        a->nframes[h] = static_cast<int>(nframes[h]);
        a->it_begin[h] = static_cast<int>(it);
        it += (nframes[h] + fpw - 1) / fpw;
        if (it > INT32_MAX) return -1;
    }
    for (int h = H; h <= kMaxHops; ++h) a->it_begin[h] = static_cast<int>(it);
    for (int h = H; h < kMaxHops; ++h) a->nframes[h] = 0;
    for (int h = 0; h < kMaxHops; ++h) a->slot_bias[h] = 0;
    const int total = static_cast<int>(it);
    const int grid = total < max_grid ? total : max_grid;
    a->total = total;
    a->q = grid ? total / grid : 0;
    a->r = grid ? total % grid : 0;
    a->step = 1;
    a->pad_ = 0;
    for (int h = 0; h <= kMaxHops; ++h) r->begin[h] = 0;
    if (grid == 0) return 0;
    if (H == 1 && interleave_single) {      // one hop: every workgroup owns exactly one slot, its own number
        a->step = grid;
        for (int h = 1; h <= kMaxHops; ++h) r->begin[h] = grid;
        return grid;
    }
    // walk the workgroups once: workgroup w touches the hops its range overlaps, in order
    int slot = 0, w = 0;
    for (int h = 0; h < H; ++h) {
        r->begin[h] = slot;
        const int b = a->it_begin[h], e = a->it_begin[h + 1];
        if (e == b) continue;
        int lo, hi;
        hop_range(w, a->q, a->r, &lo, &hi);
        while (hi <= b) hop_range(++w, a->q, a->r, &lo, &hi);      // first workgroup whose range reaches past b
        a->slot_bias[h] = slot - w;
        int wl = w;                                              // last workgroup whose range starts before e
        while (wl + 1 < grid) {
            hop_range(wl + 1, a->q, a->r, &lo, &hi);
            if (lo >= e) break;
            ++wl;
        }
        slot += wl - w + 1;
        w = wl;                                                  // it may own the next hop's first iterations too
    }
    for (int h = H; h <= kMaxHops; ++h) r->begin[h] = slot;
    return grid;
}

}  // namespace rpf
