#!/bin/bash
# Re-capture of the C4 artefacts after a change to the four-step kernels only: the bench line (both paths) and the
# rocprofv3 kernel statistics of the same command.  Output: gpurun_out/c4re/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/c4re
mkdir -p $OUT
cd $ROOT
python bench.py --workload C4 --no-end-to-end > $OUT/c4_bench.json 2> $OUT/c4_bench.err
python bench.py --workload C4 --no-end-to-end --no-cpu-baseline --engine-flags 8 > $OUT/c4_bench_two_kernel.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c4 -- python $ROOT/bench.py --workload C4 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name 'c4_kernel_stats.csv' | head -1) $OUT/c4_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cut -c1-260 $OUT/c4_bench.json; echo; cut -c1-200 $OUT/c4_bench_two_kernel.json; echo; head -4 $OUT/c4_kernel_stats.csv | cut -c1-200
