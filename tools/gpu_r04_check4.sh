#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_check4
mkdir -p $OUT
cd $ROOT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value']/1e9);print(json.dumps(d.get('end_to_end'),indent=0)[:900])"
export RPF_PARITY_RECORD=$OUT/fullsize_errors.json
rm -f $RPF_PARITY_RECORD
timeout 1200 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "float32 or thin or queue or unget or protocol" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-300; grep -n "AssertionError: (" $OUT/pytest.log | cut -c1-600
unset RPF_PARITY_RECORD
bash tools/gpu_tsan.sh 2>&1 | tail -16 | cut -c1-200
for m in 1 2; do echo "bench C4, fused mode $m"; RPF_FUSED_MODE=$m RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_fprof.so timeout 200 python bench.py --workload C4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value']/1e9, d['ms_per_step'])"; done
