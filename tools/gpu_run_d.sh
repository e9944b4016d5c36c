#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_step or golden or large_non_power" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --workload C4 --no-cpu-baseline --steps 30 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "benchc4 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2d/bench_c4.json")); print("C4: %.1f Gsample/s, %.4f ms/step, kernel %.4f ms" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
SWEEP_NOWIN=1 timeout 300 python tools/gpu_sweep.py 16384:0 65536:0 262144:0 5000:0 100000:0 2>&1 | grep "N=" | cut -c1-100
