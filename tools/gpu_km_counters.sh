#!/bin/bash
# On the GPU box: SQ counters of the planned mixed-radix kernels (how busy the VALU is), one --pmc pass with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/km_counters.csv (per-kernel means).
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/km_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SWEEP_NOWIN=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT -o km -- python $ROOT/tools/gpu_sweep.py 500:0 1000:0 5000:0 10000:0 16384:0 20000:0 4096:0 > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os, re
root = os.environ["GRAFT_REPO_ROOT"]
path = glob.glob(os.path.join(root, "gpurun_out", "km_pmc", "**", "km_counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path[0])):
    k = r["Kernel_Name"]
    if "mixed_plan_kernel" in k or "mixed_split_kernel" in k or "fft_accum_kernel" in k:
        m = re.search(r"MixPlan<(\d+)", k)
        name = ("split " if "split" in k else "KM ") + m.group(1) if m else "K1 4096"
        if int(r["Grid_Size"]) < 20000:
            continue                      # the 64-frame parity launch
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(root, "gpurun_out", "km_counters.csv"), "w") as f:
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS"]
    f.write("kernel,launches," + ",".join(cols) + "\n")
    for name, d in sorted(acc.items()):
        n = max(len(v) for v in d.values())
        f.write(name + ",%d," % n + ",".join("%.0f" % (sum(d[c]) / len(d[c])) if d.get(c) else "" for c in cols) + "\n")
print(open(os.path.join(root, "gpurun_out", "km_counters.csv")).read())
PY
rm -rf $OUT
