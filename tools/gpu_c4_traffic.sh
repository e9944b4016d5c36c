#!/bin/bash
# HBM traffic of the two four-step paths at C4's size, rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE in
# separate passes, per MI355X_MICROARCH.md).  Output: gpurun_out/c4traffic/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/gpu_fused_profile.py 262144 512"
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o c4 -- $CMD > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    files = glob.glob("$OUT/%s/**/c4_counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            key = "fused" if "fused_kernel" in n else "cols" if "cols_kernel" in n else "rows" if "rows_kernel" in n else None
            if key and r["Counter_Name"] == c:
                acc[key].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)[len(v)//2:]          # the big launches (512 frames / 128-frame batches), not the self-test
        print(c, k, "launches", len(v), "median KiB per launch %.0f" % (sorted(v)[len(v)//2]))
PY
