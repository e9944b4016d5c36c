#!/bin/bash
# SQ / LDS counters of the fused four-step kernel on config C4 (bench.py's harness), two PMC passes, no trace domains
# beside --kernel-trace.  Output: gpurun_out/c4sq/summary.txt
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/c4sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--workload C4 --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o c4 -- python $ROOT/bench.py $B > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/lds -o c4 -- python $ROOT/bench.py $B > $OUT/lds.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/c4_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "fused" if "fourstep_fused_kernel" in k else "K3" if "reduce" in k else None
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    for c, v in sorted(d.items()):
        print("%-6s %-24s launches %4d  mean %.6g" % (name, c, len(v), sum(v) / len(v)))
PY
cat $OUT/summary.txt
rm -rf $OUT/sq $OUT/lds
