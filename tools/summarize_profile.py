#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>/ (written by tools/gpu_profile.sh on the GPU box) into small
summaries under gpurun_out/<tag>/summary/ -- run ON the box, because the raw rocprofv3 output
is far beyond what gpurun copies back -- which are then copied into the tracked profiles/:

  <tag>_bench.json, <tag>_bench_20steps.json, <tag>_c3_bench.json, <tag>_c4_bench.json,
  <tag>_c5_bench.json            the bench.py lines of the same session
  <tag>_kernel_stats.csv, <tag>_c3_kernel_stats.csv, <tag>_c4_kernel_stats.csv
                                 rocprofv3 --kernel-trace --stats summaries (verbatim)
  <tag>_counters.json            per-launch means of the PMC passes (K1, K3, four-step kernels)
  traffic.json                   HBM bytes per launch / acquisition, read by bench.py
  valu_rate.json                 sustained packed-f32 issue rate (tools/lds_valu_bench), read by bench.py
  <tag>_hbm_read.txt, <tag>_lds_valu.txt, <tag>_k1_fixed_cost.txt, <tag>_sizes.txt

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read -- checked in the same run on torch's roll kernel, which reads exactly
81 920 000 B per launch."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, name):
    hits = glob.glob(os.path.join(src, sub, "**", name), recursive=True)
    return hits[0] if hits else None


def copy(path, name):
    if path and os.path.exists(path):
        shutil.copy(path, os.path.join(dst, name))


for name in ("bench.json", "bench_20steps.json", "c3_bench.json", "c4_bench.json", "c5_bench.json", "c5_rccl_bench.json",
             "c5_shard8_bench.json"):
    copy(os.path.join(src, name), tag + "_" + name)
copy(find("c2_trace", "c2_kernel_stats.csv"), tag + "_kernel_stats.csv")
copy(find("c3_trace", "c3_kernel_stats.csv"), tag + "_c3_kernel_stats.csv")
copy(find("c4_trace", "c4_kernel_stats.csv"), tag + "_c4_kernel_stats.csv")
copy(find("c5_trace", "c5_kernel_stats.csv"), tag + "_c5_kernel_stats.csv")
for name in ("hbm_read.txt", "lds_valu.txt", "k1_fixed_cost.txt", "sizes.txt", "mfma_valu.txt"):
    copy(os.path.join(src, name), tag + "_" + name)
captured = open(os.path.join(src, "captured.txt")).read().strip() if os.path.exists(os.path.join(src, "captured.txt")) else None


def classify(name):
    if "fft_accum_scan_kernel" in name:
        return "K1_scan"
    if "fft_accum_kernel" in name:
        return "K1_fft_accum"
    if "reduce_kernel" in name:
        return "K3_reduce"
    if "fourstep_fused_kernel" in name:
        return "K2f_fused"
    if "fourstep_cols_kernel" in name:
        return "K2a_cols"
    if "fourstep_rows_kernel" in name:
        return "K2b_rows"
    if "roll_cuda_kernel" in name or "roll" in name and "kernel" in name:
        return "calib_roll"
    return None


def rows_of(sub, prefix):
    path = find(sub, prefix + "_counter_collection.csv")
    return list(csv.DictReader(open(path))) if path else []


def means(rows, full_grid_only=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    k1_grids = [int(r["Grid_Size"]) for r in rows if "fft_accum_kernel" in r["Kernel_Name"]]
    full = max(k1_grids) if k1_grids else 0
    for r in rows:
        k = classify(r["Kernel_Name"])
        if not k:
            continue
        if k == "K1_fft_accum" and full_grid_only and int(r["Grid_Size"]) < full:
            continue            # the small correctness check before the timed steps
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}, \
           {k: {c: (sum(v), len(v)) for c, v in d.items()} for k, d in acc.items()}


out = {"captured": captured}
traffic = {"captured": captured,
           "source": "profiles/%s_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per "
                     "MI355X_MICROARCH.md, checked on a kernel that reads exactly 81 920 000 B)" % tag}
for cfg in ("c2", "c3"):
    merged = {}
    for sub in ("%s_pmc_FETCH_SIZE" % cfg, "%s_pmc_WRITE_SIZE" % cfg) + (("c2_pmc_sq", "c2_pmc_lds") if cfg == "c2" else ()):
        m, _ = means(rows_of(sub, cfg))
        for k, d in m.items():
            merged.setdefault(k, {}).update(d)
    out["C2" if cfg == "c2" else "C3_windowed"] = merged
    k1 = merged.get("K1_fft_accum", {})
    calib = merged.get("calib_roll", {})
    if "FETCH_SIZE" in calib and cfg == "c2":
        out["fetch_size_calibration"] = {
            "known_bytes": 81920000, "FETCH_SIZE_KiB": calib["FETCH_SIZE"],
            "bytes_per_reported_byte": 81920000.0 / (calib["FETCH_SIZE"] * 1024.0),
            "note": "MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; factor applied = 2"}
    if "FETCH_SIZE" in k1:
        traffic["fft_accum_%s_fetch_bytes_per_launch" % cfg] = k1["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in k1:
        traffic["fft_accum_%s_write_bytes_per_launch" % cfg] = k1["WRITE_SIZE"] * 1024.0
    if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
        traffic["fft_accum_%s_hbm_bytes_per_launch" % cfg] = (traffic["fft_accum_%s_fetch_bytes_per_launch" % cfg] +
                                                              traffic["fft_accum_%s_write_bytes_per_launch" % cfg])

# C5: one launch of the scan kernel = the 8 hops x 5000 frames of a scan
c5 = {}
for sub in ("c5_pmc_FETCH_SIZE", "c5_pmc_WRITE_SIZE"):
    m, _ = means(rows_of(sub, "c5"), full_grid_only=False)
    for k, d in m.items():
        c5.setdefault(k, {}).update(d)
out["C5_scan"] = c5
ks = c5.get("K1_scan", {})
if "FETCH_SIZE" in ks and "WRITE_SIZE" in ks:
    traffic["fft_accum_c5_fetch_bytes_per_launch"] = ks["FETCH_SIZE"] * 1024.0 * 2.0
    traffic["fft_accum_c5_write_bytes_per_launch"] = ks["WRITE_SIZE"] * 1024.0
    traffic["fft_accum_c5_hbm_bytes_per_launch"] = ks["FETCH_SIZE"] * 1024.0 * 2.0 + ks["WRITE_SIZE"] * 1024.0
    # the frames ONE launch of that capture covered (round 6: a launch spans several consecutive scans): bench.py scales
    # the replayed traffic to the frames of ITS launch
    try:
        c5_line = json.load(open(os.path.join(src, "c5_bench.json")))
        traffic["fft_accum_c5_frames_per_launch"] = c5_line["roofline"]["frames_per_launch"]
    except Exception:
        traffic["fft_accum_c5_frames_per_launch"] = 40000

# C4: one acquisition = ONE launch of the column kernel and one of the row kernel (round 3: 2 GB of intermediate;
# rounds 1-2: 8 batches of 128 frames)
C4_BATCHES = 1.0
c4 = {}
for sub in ("c4_pmc_FETCH_SIZE", "c4_pmc_WRITE_SIZE"):
    m, tot = means(rows_of(sub, "c4"), full_grid_only=False)
    for k, d in m.items():
        c4.setdefault(k, {}).update(d)
    for k, d in tot.items():
        for c, (s, n) in d.items():
            c4.setdefault(k + "_totals", {})[c] = {"sum_KiB": s, "launches": n}
out["C4_fourstep"] = c4
try:
    # round 4: ONE launch of the fused kernel per acquisition (+ one proof launch per engine, 16 frames, at creation)
    fz = c4["K2f_fused_totals"]
    acq = fz["FETCH_SIZE"]["launches"] - 1
    acq_w = fz["WRITE_SIZE"]["launches"] - 1
    traffic["fourstep_c4_fetch_bytes_per_launch"] = fz["FETCH_SIZE"]["sum_KiB"] * 1024.0 * 2.0 / acq
    traffic["fourstep_c4_write_bytes_per_launch"] = fz["WRITE_SIZE"]["sum_KiB"] * 1024.0 / acq_w
    traffic["fourstep_c4_hbm_bytes_per_launch"] = (traffic["fourstep_c4_fetch_bytes_per_launch"] +
                                                   traffic["fourstep_c4_write_bytes_per_launch"])
    traffic["fourstep_c4_note"] = ("per acquisition of 1000 frames = one launch of the fused four-step kernel; Y is written back once and "
                                   "about two thirds of its reads miss the L2 (two buffers of Y per team)")
except (KeyError, ZeroDivisionError):
  try:
    acq = c4["K2b_rows_totals"]["FETCH_SIZE"]["launches"] / C4_BATCHES
    fetch = (c4["K2a_cols_totals"]["FETCH_SIZE"]["sum_KiB"] + c4["K2b_rows_totals"]["FETCH_SIZE"]["sum_KiB"]) * 1024.0 * 2.0 / acq
    acq_w = c4["K2b_rows_totals"]["WRITE_SIZE"]["launches"] / C4_BATCHES
    write = (c4["K2a_cols_totals"]["WRITE_SIZE"]["sum_KiB"] + c4["K2b_rows_totals"]["WRITE_SIZE"]["sum_KiB"]) * 1024.0 / acq_w
    traffic["fourstep_c4_fetch_bytes_per_launch"] = fetch
    traffic["fourstep_c4_write_bytes_per_launch"] = write
    traffic["fourstep_c4_hbm_bytes_per_launch"] = fetch + write
    traffic["fourstep_c4_note"] = "per acquisition of 1000 frames = one launch of K2a + one of K2b; the intermediate Y is written and read once"
  except (KeyError, ZeroDivisionError):
    pass

hb = os.path.join(src, "hbm_read.txt")
if os.path.exists(hb):            # tools/hbm_read_bench.hip: read-only stream > Infinity Cache
    m = re.search(r"read-only stream of (\d+) B: .* = (\d+) GB/s", open(hb).read())
    if m:
        traffic["measured_read_only_GBps"] = float(m.group(2))
        traffic["measured_read_only_bytes"] = int(m.group(1))
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

lv = os.path.join(src, "lds_valu.txt")
if os.path.exists(lv):            # sustained packed-f32 rate: mode 0 at 4 waves per SIMD
    m = re.search(r"mode 0 .*x4/CU \(4 waves/SIMD\):\s+([\d.]+) ns per iteration", open(lv).read())
    if m:
        ns = float(m.group(1))
        json.dump({"captured": captured, "pk_fma_ns_per_instruction_per_simd": ns / 4.0 / 320.0,
                   "source": "tools/lds_valu_bench mode 0 (320 v_pk_fma_f32 per wave and iteration, 4 waves per SIMD, "
                             "every CU busy): %.1f ns per iteration" % ns,
                   "equivalent_clock_GHz_at_4_cycles_per_instruction": 4.0 / (ns / 4.0 / 320.0)},
                  open(os.path.join(dst, "valu_rate.json"), "w"), indent=1)

json.dump(out, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
