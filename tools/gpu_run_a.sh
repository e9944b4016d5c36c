#!/bin/bash
# round-2 GPU session A: full gpu tests, bench lines, K1 de-phasing experiments
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench20 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $OUT/bench_2000.json 2> $OUT/bench_2000.err; echo "bench2000 rc=$?"
timeout 300 python bench.py --workload C5 --force-dist --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "benchc5 rc=$?"
timeout 300 python bench.py --workload C4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "benchc4 rc=$?"
cut -c1-600 $OUT/bench_20.json $OUT/bench_2000.json $OUT/bench_c5.json $OUT/bench_c4.json
tail -3 $OUT/bench_*.err
export RPF_ENGINE_LIB=$GRAFT_REPO_ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_NOWIN=1
timeout 600 python tools/gpu_sweep.py 4096:0 4096:9 4096:10 4096:40 4096:41 4096:42 4096:43 4096:44 4096:45 4096:46 4096:47 2048:0 2048:40 2048:41 1024:0 1024:40 4096:0 4096:40 > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
cut -c1-210 $OUT/sweep.log
