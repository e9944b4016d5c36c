#!/bin/bash
# tools/l2_residency_bench on the GPU box: timing run, then rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
# separate runs, per MI355X_MICROARCH.md; no trace domains beside --kernel-trace).  Output: gpurun_out/l2res/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/l2res
mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/tools/l2_residency_bench
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/l2_residency_bench.hip -o $BIN
cd /tmp && export TMPDIR=/tmp
ROUNDS=${1:-1000}
timeout 120 $BIN $ROUNDS > $OUT/timing.txt 2>&1
rc=$?
cat $OUT/timing.txt
if [ $rc -ne 0 ]; then
  echo "timing run failed (rc=$rc): variants one by one, 100 rounds"
  for v in 0 1 2 3 4 5 6 7 8 9; do timeout 30 $BIN 100 $v 2>&1 | grep -v "^#" | cut -c1-200; done
  exit 1
fi
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o l2 -- $BIN $ROUNDS > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    for f in glob.glob("$OUT/%s/**/l2_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c or not ("residency_kernel" in r["Kernel_Name"] or "team32_kernel" in r["Kernel_Name"]):
                continue
            m = re.search(r"((?:residency|team32)_kernel<[^>]*>)", r["Kernel_Name"])
            rows.setdefault(m.group(1), collections.defaultdict(list))[c].append(float(r["Counter_Value"]))
print("# per launch, in launch order (2 launches per size: 1, 2, 3 MB per team); FETCH/WRITE_SIZE in KiB as reported")
print("# template args = <store kind, load kind, foreign, cross>: store 0 plain 1 nt 2 sc1 3 sc0sc1; load 0 inv+plain 1 sc1 2 plain-noinv 3 sc0sc1")
for k, d in rows.items():
    for c, v in d.items():
        print("%-28s %-12s %s" % (k, c, " ".join("%.0f" % x for x in v)))
PY
