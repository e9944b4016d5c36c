#!/usr/bin/env python3
"""Candidate plans for the planned mixed-radix kernel (rpf_mixed.hip / mixed_core.h).

  python tools/gen_mixed_plans.py splitsearch 20000 50000 ... >> rtl-power-fftw_amd/csrc/mixed_plans_tuning.inc
      candidates of the split form (N = P x M) for the listed sizes, variants 11, 12, ...; `splitcases` prints the
      N:variant arguments for tools/gpu_sweep.py.  (mixed_plans_split.inc itself is hand-listed.)
  python tools/gen_mixed_plans.py search  > rtl-power-fftw_amd/csrc/mixed_plans_tuning.inc
      every candidate (radices, butterflies per thread, frame slots per workgroup, twiddle
      placement) of every size in SIZES as a variant of the tuning build; tools/gpu_sweep.py
      times them on the GPU (N:variant), tools/pick_mixed_plans.py turns the log into
      mixed_plans.inc -- the table the shipped library is built with.

A plan is N = R_0 ... R_{F-1} with G_i butterflies per thread in pass i (R_i G_i points per thread,
TPF_i = N / (R_i G_i) threads of a frame take part), FPW frame slots per workgroup and TW = 0
(twiddles in registers) or 1 (per-thread rows of an LDS table)."""
import itertools
import os
import sys

RADICES = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25]
# packed-f32 instructions of one butterfly (counted from dft_small.h)
COST = {2: 2, 3: 8, 4: 8, 5: 20, 6: 22, 8: 28, 9: 56, 10: 50, 12: 56, 15: 100, 16: 80, 18: 130, 20: 120,
        24: 148, 25: 232, 7: 33, 11: 75, 13: 102, 17: 170, 19: 210, 23: 300, 14: 80, 21: 155, 22: 172}
# the decimal "round" sizes people type, and the 3- and 5-multiples of powers of two
SIZES = [100, 120, 150, 160, 200, 240, 250, 300, 320, 360, 400, 450, 480, 500, 600, 640, 720, 750, 800, 900, 960, 1000,
         1200, 1250, 1280, 1440, 1500, 1600, 1800, 1920, 2000, 2160, 2400, 2500, 2560, 2700, 2880, 3000, 3200, 3600, 3750,
         3840, 4000, 4500, 4800, 5000, 5120, 5400, 5760, 6000, 6250, 6400, 7200, 7500, 7680, 8000, 8100, 8640, 9000, 9600,
         10000, 150, 750, 3750, 6250,
         # the remaining multiples of 10 with prime factors 2, 3, 5, and 3 2^k / 9 2^k
         50, 60, 80, 90, 180, 270, 540, 810, 1080, 1350, 1620, 2250, 2430, 3240, 4050, 4320, 4860, 6480, 6750, 7290, 9720,
         96, 192, 384, 768, 1536, 3072, 6144, 576, 1152, 2304, 4608, 9216,
         # multiples of 100 with a prime factor 7 ... 23 (dft_small.h's PrimeDft)
         140, 350, 700, 1100, 1300, 1400, 1700, 1900, 2100, 2200, 2300, 2600, 2800, 3300, 3400, 3500, 3800, 3900, 4200, 4400,
         4600, 4900, 5100, 5200, 5500, 5600, 5700, 6300, 6500, 6600, 6800, 6900, 7000, 7600, 7700, 7800, 8400, 8500, 8800,
         9100, 9200, 9500, 9800, 9900]
LDS_LIMIT = 160 * 1024
TW_MODES = (0, 1)        # the sizes searched so far were searched with these; "search2" adds mode 2
MAXPPT = 25              # ... and up to 32 points per thread


def factorisations(n, maxf=4):
    out = []

    def rec(rem, cur):
        if rem == 1:
            out.append(tuple(cur))
            return
        if len(cur) == maxf:
            return
        for r in RADICES:
            if rem % r == 0:
                rec(rem // r, cur + [r])
    rec(n, [])
    return out


def plans_of(n, ratio=0.74):
    res = []
    for rad in factorisations(n):
        f = len(rad)
        if f < 2:
            continue
        opts = [[g for g in range(1, 9) if 8 <= r * g <= MAXPPT and n % (r * g) == 0] or
                [g for g in range(1, 9) if r * g <= MAXPPT and n % (r * g) == 0][-1:] for r in rad]
        if any(not o for o in opts):
            continue
        for gs in itertools.product(*opts):
            tpf = [n // (r * g) for r, g in zip(rad, gs)]
            tmax = max(tpf)
            if min(tpf) < ratio * tmax:
                continue
            per = 0.0
            for i, (r, g) in enumerate(zip(rad, gs)):
                last = i == f - 1
                per += g * (COST[r] + (4 * r if last else 2 * (r - 1)) + (1 if (i == 0 or last) else 2) * r) + 40
            per += 3 * rad[0] * gs[0]
            res.append((per * tmax / n, rad, gs, tpf, tmax))
    res.sort()
    return res


def lds_bytes(n, rad, gs, fpw, tw):
    rlast = rad[-1]
    cpx = n + n // rlast if rlast % 2 == 0 else n
    table = 0
    if tw == 1:
        table = sum(g * (r - 1) * (n // (r * g)) for r, g in zip(rad[:-1], gs[:-1]))
    elif tw == 2:
        s = n
        for i, r in enumerate(rad[:-1]):
            s //= r                       # S_i
            if i >= 1:
                table += (r - 1) * s
    return (fpw * cpx + table) * 8


def vgprs(rad, gs, tw):
    pts = 2 * max(r * g for r, g in zip(rad, gs))
    twr = {0: 2 * sum(g * (r - 1) for r, g in zip(rad[:-1], gs[:-1])), 1: 0, 2: 2 * gs[0] * (rad[0] - 1)}[tw]
    return pts + twr + 2 * rad[-1] * gs[-1] + (rad[0] * gs[0] + 1) // 2 + 24


SINGLE_SLOT = False      # split form: one frame slot per workgroup whatever the length (several workgroups share a CU)


def candidates(n, per_size=18):
    out = _candidates(n, per_size, 0.74) or _candidates(n, per_size, 0.66) or _candidates(n, per_size, 0.5)
    for rad, gs, fpw, tw in EXTRA.get(n, []):
        prod = 1
        for r, g in zip(rad, gs):
            prod *= r
            assert n % (r * g) == 0, (n, rad, gs)
        assert prod == n and fpw * max(n // (r * g) for r, g in zip(rad, gs)) <= 1024, (n, rad, gs, fpw)
        out.append((0.0, rad, gs, fpw, tw))
    return out


def _candidates(n, per_size, ratio):
    out = []
    plans = plans_of(n, ratio)
    for lo, hi, nshapes in ((0, 12, 2), (13, 16, 1), (17, 25, 2), (26, 32, 2)):       # points per thread: light / middle / heavy
        seen_shapes = set()
        for cost, rad, gs, tpf, tmax in plans:
            ppt = max(r * g for r, g in zip(rad, gs))
            if not lo <= ppt <= hi:
                continue
            shape = (tuple(sorted(zip(rad, gs))), rad[-1])
            if shape in seen_shapes:
                continue
            fpws = []
            for fpw in range(1, 65):
                wg = fpw * tmax
                if wg > 1024:
                    break
                waves = -(-wg // 64)
                if wg < 120 or wg / (64.0 * waves) < 0.88:
                    continue
                fpws.append(fpw)
            pick = []
            for target in (256, 512):
                best = min(fpws, key=lambda f: abs(f * tmax - target), default=None)
                if best is not None and best not in pick:
                    pick.append(best)
            if SINGLE_SLOT:
                pick = [1] if 60 <= tmax <= 1024 and tmax / (64.0 * -(-tmax // 64)) >= 0.8 else []
            got = False
            for fpw in pick:
                for tw in TW_MODES:
                    if tw == 2 and len(rad) < 3:
                        continue          # (two passes: the same as registers)
                    wg = fpw * tmax
                    v = vgprs(rad, gs, tw)
                    waves_per_simd = -(-wg // 64) / 4.0
                    if v * max(1.0, waves_per_simd) > 512:
                        continue
                    if lds_bytes(n, rad, gs, fpw, tw) > LDS_LIMIT:
                        continue
                    out.append((cost, rad, gs, fpw, tw))
                    got = True
            if got:
                seen_shapes.add(shape)
            if len(seen_shapes) >= nshapes:
                break
    return out[:per_size]


# hand-added candidates the filters above reject (register estimate too cautious)
EXTRA = {8192: [((16, 8, 8, 8), (2, 4, 4, 4), 1, 2), ((16, 8, 8, 8), (2, 4, 4, 4), 2, 2), ((8, 8, 8, 16), (4, 4, 4, 2), 1, 2),
                ((8, 8, 8, 16), (4, 4, 4, 2), 2, 0), ((8, 16, 8, 8), (4, 2, 4, 4), 1, 0), ((4, 8, 16, 16), (8, 4, 2, 2), 1, 2)],
         16384: [((4, 16, 16, 16), (4, 1, 1, 1), 1, 2), ((8, 8, 16, 16), (2, 2, 1, 1), 1, 2), ((8, 16, 8, 16), (2, 1, 2, 1), 1, 2),
                 ((16, 16, 4, 16), (1, 1, 4, 1), 1, 2), ((8, 16, 8, 16), (4, 2, 4, 2), 1, 2),
                 ((16, 4, 16, 16), (2, 8, 2, 2), 1, 2), ((8, 8, 16, 16), (4, 4, 2, 2), 1, 2), ((16, 16, 4, 16), (2, 2, 8, 2), 1, 2),
                 ((16, 8, 8, 16), (2, 4, 4, 2), 1, 2)],
         9000: [((10, 10, 10, 9), (1, 1, 1, 1), 1, 0), ((9, 10, 10, 10), (1, 1, 1, 1), 1, 0),
                ((10, 9, 10, 10), (1, 1, 1, 1), 1, 0), ((18, 20, 25), (1, 1, 1), 1, 0), ((25, 18, 20), (1, 1, 1), 1, 0)],
         6250: [((10, 25, 25), (1, 1, 1), 1, 0), ((25, 10, 25), (1, 1, 1), 1, 0), ((25, 25, 10), (1, 1, 1), 1, 0)],
         3750: [((10, 15, 25), (3, 2, 1), 2, 0), ((10, 15, 25), (3, 2, 1), 3, 0), ((6, 25, 25), (5, 1, 1), 2, 0),
                ((6, 25, 25), (5, 1, 1), 3, 0), ((25, 25, 6), (1, 1, 5), 2, 0), ((6, 25, 25), (1, 1, 1), 1, 0)],
         750: [((15, 10, 5), (1, 1, 2), 5, 0), ((10, 15, 5), (1, 1, 3), 5, 0), ((25, 6, 5), (1, 5, 5), 8, 0),
               ((5, 10, 15), (2, 1, 1), 5, 0), ((30 // 2, 10, 5), (1, 1, 2), 3, 0)],
         150: [((15, 10), (1, 1), 16, 0), ((10, 15), (1, 1), 16, 0), ((15, 10), (1, 1), 32, 0), ((6, 25), (5, 1), 40, 0),
               ((25, 6), (1, 5), 40, 0)]}


def split_candidates(n):
    """The split form (mixed_split_kernel): n = P x M, P = 2 ... 5 (6, 8, 10: its paired form), M <= 16384 even -- every single-slot candidate of M
    (search2 rules) with its later passes' twiddles in LDS tables (mode 2, what the split kernel's registers allow)."""
    global TW_MODES, MAXPPT, SINGLE_SLOT
    saved = TW_MODES, MAXPPT, SINGLE_SLOT
    TW_MODES, MAXPPT = (0, 1, 2), 32
    out, seen = [], set()
    try:
        for p in (2, 3, 4, 5, 6, 8, 10):     # (6, 8, 10: the paired form)
            m = n // p
            if n % p or m % 2 or m > 16384:
                continue
            SINGLE_SLOT = m < 4000      # (from 4000 up the plain search's own candidates are single-slot)
            for cost, rad, gs, fpw, tw in candidates(m):
                key = (p, rad, gs)
                if fpw != 1 or tw == 1 or len(rad) < 3 or key in seen:
                    continue
                if lds_bytes(m, rad, gs, 1, 2) > LDS_LIMIT:       # (a hand-added candidate whose mode-2 table does not fit)
                    continue
                seen.add(key)
                out.append((p, m, rad, gs))
    finally:
        TW_MODES, MAXPPT, SINGLE_SLOT = saved
    return out


def split_entry(p, m, rad, gs, variant, wm=0):
    passes = ", ".join("P<%d%s>" % (r, (", %d" % g) if g != 1 else "") for r, g in zip(rad, gs))
    return "    split_entry<%d, MixPlan<%d, 1, 2, %s>%s>(%d)," % (p, m, passes, (", %d" % wm) if wm else "", variant)


def measured_split_rates():
    """(N, P, M, radices, groups) -> Gsample/s of the plain split form, from the committed search results."""
    import ast, re
    rates, n = {}, None
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for name in ("r02_split_plan_search.txt", "r03_split_plan_search.txt"):
        path = os.path.join(prof, name)
        if not os.path.exists(path):
            continue
        for line in open(path):
            m = re.match(r"N=(\d+)", line)
            if m:
                n = int(m.group(1))
                continue
            m = re.match(r"\s+([\d.]+)\s+\(\S+\)\s+P (\d) M (\d+) (\([\d, ]+\)) (\([\d, ]+\))", line)
            if m and n:
                rates[(n, int(m.group(2)), int(m.group(3)), ast.literal_eval(m.group(4)), ast.literal_eval(m.group(5)))] = float(m.group(1))
    return rates


def window_candidates(n, keep=6):
    """Windowed twins of the split form for size n: (P, M, radices, groups, window mode) -- mode 2 (window values
    fetched a section ahead) for the `keep` fastest plain candidates, mode 3 (the whole window in LDS) for the `keep`
    fastest candidates that have the room.  Sizes the plain search did not cover (16384) keep every candidate."""
    rates = measured_split_rates()
    cands = split_candidates(n)
    known = [c for c in cands if (n,) + c in rates]
    if known:
        cands = sorted(known, key=lambda c: -rates[(n,) + c])
    fits = [c for c in cands if lds_bytes(c[1], c[2], c[3], 1, 2) + 4 * n <= LDS_LIMIT]
    lim = keep if known else len(cands)
    return [c + (2,) for c in cands[:lim]] + [c + (3,) for c in fits[:lim]]


def variant_base(n):
    """K1's own tuning variants use the low numbers for the powers of two."""
    return 100 if n & (n - 1) == 0 else 0


def entry(n, rad, gs, fpw, tw, variant):
    passes = ", ".join("P<%d%s>" % (r, (", %d" % g) if g != 1 else "") for r, g in zip(rad, gs))
    return "    plan_candidate<MixPlan<%d, %d, %d, %s>>(%d)," % (n, fpw, tw, passes, variant)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "search"
    sizes = [int(a) for a in sys.argv[2:]] or SIZES
    if mode.endswith("2"):
        global TW_MODES, MAXPPT
        TW_MODES = (0, 1, 2)
        MAXPPT = 32
        mode = mode[:-1]
    if mode == "splitsearch":      # candidates of the split form as variants 11, 12, ... of the tuning build
        print("// generated by tools/gen_mixed_plans.py splitsearch -- tuning build only")
        for n in sizes:
            for v, (p, m, rad, gs) in enumerate(split_candidates(n), start=11 + variant_base(n)):
                print(split_entry(p, m, rad, gs, v))
        return
    if mode == "splitcases":
        print(" ".join("%d:%d" % (n, v + variant_base(n)) for n in sizes for v in range(11, 11 + len(split_candidates(n)))))
        return
    if mode == "winsearch":        # windowed twins of split candidates, variants 11, 12, ... (tuning build)
        print("// generated by tools/gen_mixed_plans.py winsearch -- tuning build only")
        for n in sizes:
            for v, (p, m, rad, gs, wm) in enumerate(window_candidates(n), start=11 + variant_base(n)):
                print(split_entry(p, m, rad, gs, v, wm))
        return
    if mode == "wincases":
        print(" ".join("%d:%d" % (n, v + variant_base(n)) for n in sizes for v in range(11, 11 + len(window_candidates(n)))))
        return
    if mode == "winlist":
        for n in sizes:
            for v, (p, m, rad, gs, wm) in enumerate(window_candidates(n), start=11 + variant_base(n)):
                print(n, v, "P", p, "M", m, rad, gs, "WM", wm)
        return
    if mode == "splitlist":
        for n in sizes:
            for v, (p, m, rad, gs) in enumerate(split_candidates(n), start=11 + variant_base(n)):
                print(n, v, "P", p, "M", m, rad, gs)
        return
    if mode == "search":
        print("// generated by tools/gen_mixed_plans.py search -- tuning build only")
        for n in sizes:
            for v, (cost, rad, gs, fpw, tw) in enumerate(candidates(n), start=1 + variant_base(n)):
                print(entry(n, rad, gs, fpw, tw, v))
    elif mode == "cases":          # the N:variant arguments for tools/gpu_sweep.py
        print(" ".join("%d:%d" % (n, v + variant_base(n)) for n in sizes for v in range(1, len(candidates(n)) + 1)))
    elif mode == "list":
        for n in sizes:
            for v, (cost, rad, gs, fpw, tw) in enumerate(candidates(n), start=1):
                print(n, v, "cost %.1f" % cost, rad, gs, "fpw", fpw, "tw", tw, "wg", fpw * max(n // (r * g) for r, g in zip(rad, gs)),
                      "lds", lds_bytes(n, rad, gs, fpw, tw))


if __name__ == "__main__":
    main()
