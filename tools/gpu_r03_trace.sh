#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03trace
mkdir -p $OUT
export TMPDIR=/tmp
for tree in old new; do
  D=$ROOT; [ $tree = old ] && D=$ROOT/.ab_r02
  for R in 512 10000; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tree}_$R -o x -- python $D/tools/gpu_k1_loop.py $R 500 > $OUT/${tree}_$R.log 2>&1) || true
    f=$(find $OUT/${tree}_$R -name "*kernel_stats.csv" | head -1)
    echo "== $tree R=$R"; grep "plain\|per launch" $OUT/${tree}_$R.log | tail -1; python3 -c "import csv,sys; [print(r[\"Name\"][:60], r[\"Calls\"], r[\"AverageNs\"], r[\"MinNs\"]) for r in csv.DictReader(open(sys.argv[1])) if \"fft_accum\" in r[\"Name\"]]" "$f"
  done
done
rm -rf $OUT/old_* $OUT/new_*
