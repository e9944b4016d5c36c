#!/bin/bash
# C4 in bench.py's harness, fused kernel (shipped) against the two-kernel path on the same box; the fused parity tests;
# the profile build's per-segment times.
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload C4 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms, kernel', round(d['roofline']['kernel_ms'],4))"; }
L=$GRAFT_REPO_ROOT/rtl-power-fftw_amd
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_fused or four_step" 2>&1 | tail -2
for rep in 1 2; do
echo "fused (shipped): $(run)"
echo "two-kernel     : $(run --engine-flags 8)"
done
RPF_FUSED_MODE=1 RPF_ENGINE_LIB=$L/librpf_engine_fprof.so timeout 150 python tools/gpu_fused_profile.py 262144 1000 2>&1 | grep -E "^fused|total|   [PC] |team 0:" | cut -c1-220
