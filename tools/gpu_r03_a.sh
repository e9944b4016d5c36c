#!/bin/bash
# round 3, first GPU pass: the multi-hop launch -- parity, C2/C5 bench lines, fixed costs, raw-ring depth
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
date -u +%Y-%m-%dT%H:%M:%SZ > $OUT/captured.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_20steps.json 2> $OUT/bench.err
timeout 300 python bench.py --workload C5 --no-cpu-baseline --no-end-to-end > $OUT/c5_bench.json 2>> $OUT/bench.err
timeout 300 python bench.py --workload C5 --shard-as 8 --no-cpu-baseline --no-end-to-end > $OUT/c5_shard8.json 2>> $OUT/bench.err
cut -c1-300 $OUT/bench_20steps.json $OUT/c5_bench.json $OUT/c5_shard8.json
timeout 300 python tools/gpu_fixed_cost.py > $OUT/k1_fixed_cost.txt 2>&1; cat $OUT/k1_fixed_cost.txt
RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_NOWIN=1 timeout 300 python tools/gpu_sweep.py 4096:0 4096:50 4096:51 4096:52 4096:53 4096:54 2048:0 2048:51 > $OUT/rawd.txt 2>&1; cat $OUT/rawd.txt
