#!/bin/bash
# C4 in bench.py's harness: the tree's library against a copy of the previous one (librpf_engine_prev.so), same box; the
# fused parity tests first.
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload C4 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms, kernel', round(d['roofline']['kernel_ms'],4))"; }
L=$GRAFT_REPO_ROOT/rtl-power-fftw_amd
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_fused or four_step" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
echo "fused (tree)    : $(run)"
echo "fused (previous): $(RPF_ENGINE_LIB=$L/librpf_engine_prev.so run)"
done
