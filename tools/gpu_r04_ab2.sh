#!/bin/bash
# C4 in bench.py's harness: fused kernel with two buffers of Y per team (shipped) / one buffer (librpf_engine_nbuf1.so) /
# the two-kernel path, same box.
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload C4 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms, kernel', round(d['roofline']['kernel_ms'],4))"; }
L=$GRAFT_REPO_ROOT/rtl-power-fftw_amd
for rep in 1 2; do
echo "fused, two buffers (shipped): $(run)"
echo "fused, one buffer           : $(RPF_ENGINE_LIB=$L/librpf_engine_nbuf1.so run)"
echo "two-kernel                         : $(run --engine-flags 8)"
done
for lib in librpf_engine.so librpf_engine_nbuf1.so; do echo $lib; RPF_ENGINE_LIB=$GRAFT_REPO_ROOT/rtl-power-fftw_amd/$lib timeout 100 python tools/gpu_fused_profile.py 262144 1000 2>&1 | grep -E "^fused|^two"; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_fused or four_step" 2>&1 | tail -2
RPF_ENGINE_LIB=$L/librpf_engine_nbuf1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_fused or four_step" 2>&1 | tail -2
