#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_20steps.json 2> $OUT/bench.err
timeout 300 python bench.py --workload C5 --no-cpu-baseline --no-end-to-end > $OUT/c5_bench.json 2>> $OUT/bench.err
timeout 300 python bench.py --workload C5 --shard-as 8 --no-cpu-baseline --no-end-to-end > $OUT/c5_shard8.json 2>> $OUT/bench.err
python - <<'PY'
import json
for f in ("bench_20steps","c5_bench","c5_shard8"):
    d=json.load(open("gpurun_out/r03b/%s.json"%f)); r=d["roofline"]
    print(f, "value %.4g ms_per_step %.5f kernel_ms %.5f frac %.3f"%(d["value"],d["ms_per_step"],r["kernel_ms"],r["frac"]))
PY
timeout 300 python tools/gpu_fixed_cost.py > $OUT/k1_fixed_cost.txt 2>&1; cat $OUT/k1_fixed_cost.txt
