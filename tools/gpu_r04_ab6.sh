#!/bin/bash
# The tree's library against a copy of the previous one (librpf_engine_prev.so), same box: fused parity tests, C4 in
# bench.py's harness, and the smaller fused sizes kernel-only (82 MB launches, rectangular).
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload C4 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms, kernel', round(d['roofline']['kernel_ms'],4))"; }
L=$GRAFT_REPO_ROOT/rtl-power-fftw_amd
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_fused or four_step" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
echo "fused (tree)    : $(run)"
echo "fused (previous): $(RPF_ENGINE_LIB=$L/librpf_engine_prev.so run)"
done
for lib in librpf_engine.so librpf_engine_prev.so; do echo $lib; RPF_ENGINE_LIB=$L/$lib SWEEP_NOWIN=1 SWEEP_K=100 SWEEP_FLAGS=2 timeout 200 python tools/gpu_sweep.py 16384:0 32768:0 65536:0 131072:0 2>&1 | grep Gsample | cut -c1-60; done
