#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_check2
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/gpu_heldout_alternatives.py 52000 64000 72000 75000 76000 77000 90000 98304 100000 105000 262144 2>&1 | grep -v amdgpu.ids | tee $OUT/heldout_alternatives.txt | cut -c1-330
for f in 0 8; do timeout 300 python bench.py --workload C4 --steps 20 --warmup 5 --engine-flags $f --no-cpu-baseline > $OUT/bench_c4_flags$f.json 2>/dev/null; echo "bench C4 flags=$f: $(python3 -c "import json;d=json.load(open('$OUT/bench_c4_flags$f.json'));print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'])")"; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host.py tests/test_hops.py -m gpu -q -x -k "queue or unget or protocol or get_power or cli or golden" 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value']/1e9);print(json.dumps(d.get('end_to_end'),indent=0)[:900])"
bash tools/gpu_tsan.sh 2>&1 | tail -40 | cut -c1-250
