#!/bin/bash
# Round 4 mid-round check on the GPU box: the whole -m gpu suite (no -x: every failure listed) with the parity record,
# C4 / default bench lines, then the ThreadSanitizer pass.  Output: gpurun_out/r04_check/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_check
mkdir -p $OUT
cd $ROOT
rm -f $OUT/fullsize_errors.json
export RPF_PARITY_RECORD=$OUT/fullsize_errors.json
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-250
unset RPF_PARITY_RECORD
timeout 300 python bench.py --workload C4 --steps 20 --warmup 5 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench C4 rc=$?"; cut -c1-600 $OUT/bench_c4.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
bash tools/gpu_tsan.sh 2>&1 | tail -60 | cut -c1-250
