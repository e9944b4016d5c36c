#!/usr/bin/env python3
"""Rank the split-form candidates of a GPU sweep and print the table lines of the picks.
  python tools/pick_split_plans.py rank  <sweep output> sizes ...   > profiles/rNN_split_plan_search.txt  (ranking, plain runs)
  python tools/pick_split_plans.py plain <sweep output> sizes ...   split_entry lines for mixed_plans_split.inc
  python tools/pick_split_plans.py windowed <sweep output of wincases> <shipped sweep> sizes ...   override lines
  python tools/pick_split_plans.py both <sweep output, plain and windowed> <shipped sweep> sizes ...
      plain and windowed picks from one sweep of the splitsearch candidates (each with its default windowed twin) against
      what ships: table lines for new sizes, override lines where a shipped size gains 5 % or more
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
    if mode == "both":
        shipped = {(n, w): (rate, err) for n, v, w, rate, err in parse(sys.argv[3]) if v == 0}
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rtl-power-fftw_amd", "csrc")
        in_table = set(int(m) for m in re.findall(r"split_entry<.*\(0\),\s+// (\d+)", open(os.path.join(csrc, "mixed_plans_split.inc")).read()))
        table, overrides = [], []
        for n in [int(a) for a in sys.argv[4:]]:
            cands = g.split_candidates(n)
            got = {0: [], 1: []}
            for v, (p, m, rad, gs) in enumerate(cands, start=11 + g.variant_base(n)):
                for w in (0, 1):
                    hit = [(rate, err) for nn, vv, ww, rate, err in rows if nn == n and vv == v and ww == w]
                    if hit:
                        got[w].append((hit[0][0], hit[0][1], p, m, rad, gs))
            pick = {w: choose(sorted(got[w], reverse=True)) for w in (0, 1)}
            if pick[0] is None:
                print("// %d: no plain candidate within %.1e" % (n, FALLBACK), file=sys.stderr)
                continue
            def wm_of(c):      # the default windowed twin of split_entry
                return 2 if c[2] > 5 else 3 if g.lds_bytes(c[3], c[4], c[5], 1, 2) + 4 * n <= g.LDS_LIMIT else 1
            new_size = n not in in_table and shipped.get((n, 0), (0, 0))[0] < 0.6 * pick[0][0]
            if new_size:
                rate, err, p, m, rad, gs = pick[0]
                table.append("    split_entry<%d, MixPlan<%d, 1, 2, %s>>(0),   // %d   %.0f (%.1e) <- %.0f"
                             % (p, m, passes(rad, gs), n, rate, err, shipped.get((n, 0), (0, 0))[0]))
                if pick[1] and pick[1][2:] != pick[0][2:]:
                    twin = [c for c in got[1] if c[2:] == pick[0][2:]]
                    if not twin or pick[1][0] > 1.03 * twin[0][0] or twin[0][1] > FALLBACK:
                        c = pick[1]
                        overrides.append("    {%d, true, split_form<%d, MixPlan<%d, 1, 2, %s>, %d>()},   // %.0f (%.1e) <- %.0f"
                                         % (n, c[2], c[3], passes(c[4], c[5]), wm_of(c), c[0], c[1], twin[0][0] if twin else 0))
            else:
                for w in (0, 1):
                    c, old = pick[w], shipped.get((n, w))
                    if c and old and c[0] > 1.05 * old[0]:
                        overrides.append("    {%d, %s, split_form<%d, MixPlan<%d, 1, 2, %s>, %d>()},   // %.0f (%.1e) <- %.0f (%.1e)"
                                         % (n, "true" if w else "false", c[2], c[3], passes(c[4], c[5]), wm_of(c) if w else 0, c[0], c[1], old[0], old[1]))
        print("// table lines (mixed_plans_split.inc)")
        print("\n".join(table))
        print("// override lines (mixed_plans_override.inc)")
        print("\n".join(overrides))
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
