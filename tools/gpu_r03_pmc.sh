#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03pmc
mkdir -p $OUT
export TMPDIR=/tmp
for tree in old new; do
  D=$ROOT; [ $tree = old ] && D=$ROOT/.ab_r02
  cd $D
  python tools/gpu_k1_loop.py 10000 2000 > $OUT/${tree}_plain.txt 2>&1
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/${tree}_pmc$i -o x -- python $D/tools/gpu_k1_loop.py 10000 300 > $OUT/${tree}_pmc$i.log 2>&1) || true
  done
done
cd $ROOT
python - <<'PY'
import csv, glob, collections, os
out = "/root/repo/gpurun_out/r03pmc"
for tree in ("old", "new"):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + "/%s_pmc*/**/*counter_collection.csv" % tree, recursive=True):
        for row in csv.DictReader(open(f)):
            if "fft_accum" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(tree, open(out + "/%s_plain.txt" % tree).read().strip().splitlines()[-1])
    for k in sorted(agg):
        v = agg[k]; print("   %-24s mean %14.1f  n=%d" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/*_pmc[0-9]
