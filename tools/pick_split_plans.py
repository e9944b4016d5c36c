#!/usr/bin/env python3
"""Rank the split-form candidates of a GPU sweep and print the table lines of the picks.
  python tools/pick_split_plans.py rank  <sweep output> sizes ...   > profiles/rNN_split_plan_search.txt  (ranking, plain runs)
  python tools/pick_split_plans.py plain <sweep output> sizes ...   split_entry lines for mixed_plans_split.inc
  python tools/pick_split_plans.py windowed <sweep output of wincases> <shipped sweep> sizes ...   override lines
The sweep output is tools/gpu_sweep.py's over `gen_mixed_plans.py splitcases` (resp. `wincases`) of the same sizes.
Pick: the fastest candidate within LIMIT (5e-7) of float64 truth -- unless one within SOFT (5.5e-7) is 1.5 x faster; no
candidate within LIMIT: the fastest within FALLBACK (6.5e-7, where the shipped table's least accurate sizes sit), else
the size is left out.  What decides in the end is the parity test on the tone stream (GPU against the CPU path < 1e-6)."""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_mixed_plans as g

LIMIT, SOFT, FALLBACK = 5e-7, 5.5e-7, 6.5e-7


def choose(got):
    """got: [(rate, err, ...)] sorted by rate, fastest first"""
    ok = [c for c in got if c[1] <= LIMIT]
    soft = [c for c in got if c[1] <= SOFT]
    if ok and soft and soft[0][0] >= 1.5 * ok[0][0]:
        return soft[0]
    if ok:
        return ok[0]
    fb = [c for c in got if c[1] <= FALLBACK]
    return fb[0] if fb else None


def parse(path):
    out = []
    for line in open(path):
        m = re.match(r"N=\s*(\d+) v=(\d+) win=(\d)\s+K1 ([\d.]+) ms\s+([\d.]+) Gsample/s.*err-vs-f64 (\S+)", line)
        if m:
            out.append((int(m[1]), int(m[2]), int(m[3]), float(m[5]), float(m[6])))
    return out


def passes(rad, gs):
    return ", ".join("P<%d%s>" % (r, (", %d" % k) if k != 1 else "") for r, k in zip(rad, gs))


def main():
    mode, path = sys.argv[1], sys.argv[2]
    rows = parse(path)
    if mode == "windowed":
        shipped = {(n, w): (rate, err) for n, v, w, rate, err in parse(sys.argv[3]) if v == 0}
        sizes = [int(a) for a in sys.argv[4:]]
        for n in sizes:
            cands = g.window_candidates(n)
            got = []
            for v, (p, m, rad, gs, wm) in enumerate(cands, start=11 + g.variant_base(n)):
                hit = [(rate, err) for nn, vv, w, rate, err in rows if nn == n and vv == v and w == 1]
                if hit:
                    got.append((hit[0][0], hit[0][1], p, m, rad, gs, wm))
            got.sort(reverse=True)
            best = choose(got)
            twin = shipped.get((n, 1))
            plain = shipped.get((n, 0))
            if best and twin and best[0] > 1.03 * twin[0]:
                print("    {%d, true, split_form<%d, MixPlan<%d, 1, 2, %s>, %d>()},   // %.0f (%.0f %%) <- %.0f (%.0f %%)"
                      % (n, best[2], best[3], passes(best[4], best[5]), best[6], best[0], 100 * best[0] / plain[0], twin[0],
                         100 * twin[0] / plain[0]))
        return
    sizes = [int(a) for a in sys.argv[3:]]
    if mode == "rank":
        print("Split form of the mixed-radix kernel (N = P x M): every candidate of tools/gen_mixed_plans.py splitsearch timed on one")
        print("MI355X, no window, 81.92 MB per launch; Gsample/s (error vs float64 truth, 64 frames); the table ships the fastest")
        print("within 5e-7 (tools/pick_split_plans.py).")
    for n in sizes:
        cands = g.split_candidates(n)
        got = []
        for v, (p, m, rad, gs) in enumerate(cands, start=11 + g.variant_base(n)):
            hit = [(rate, err) for nn, vv, w, rate, err in rows if nn == n and vv == v and w == 0]
            if hit:
                got.append((hit[0][0], hit[0][1], p, m, rad, gs))
        got.sort(reverse=True)
        if mode == "rank":
            print("N=%d" % n)
            for rate, err, p, m, rad, gs in got:
                print("    %5.1f  (%.1e)  P %d M %d %s %s" % (rate, err, p, m, rad, gs))
        else:
            pick = choose(got)
            if pick:
                rate, err, p, m, rad, gs = pick
                print("    split_entry<%d, MixPlan<%d, 1, 2, %s>>(0),   // %d   %.0f (%.1e)" % (p, m, passes(rad, gs), n, rate, err))
            else:
                print("// %d: no candidate within %.1e (best %s)" % (n, FALLBACK, "%.0f at %.1e" % got[0][:2] if got else "none"), file=sys.stderr)


if __name__ == "__main__":
    main()
