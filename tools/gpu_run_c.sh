#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_step or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 300 python bench.py --workload C4 --no-cpu-baseline --steps 30 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "benchc4 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c/bench_c4.json")); print("C4 fused: %.1f Gsample/s, %.4f ms/step, kernel %.4f ms" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
