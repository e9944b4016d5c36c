#!/bin/bash
# On the GPU box: every shipped split form beside its one-buffer ("rolling") twin of the tuning build
# (tools/gen_roll_candidates.py), plain and windowed; then the parity tests of the sizes with overrides.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/roll
export SWEEP_K=${SWEEP_K:-120}
CASES=$(python tools/gen_roll_candidates.py cases)
SHIPPED=$(for c in $CASES; do echo "${c%%:*}:0"; done | sort -u -t: -k1,1n | tr '\n' ' ')
timeout 600 python tools/gpu_sweep.py $SHIPPED > gpurun_out/roll/shipped.txt 2>&1
RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so timeout 900 python tools/gpu_sweep.py $CASES > gpurun_out/roll/search.txt 2>&1
grep -c Gsample gpurun_out/roll/shipped.txt gpurun_out/roll/search.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_radix or thin" > gpurun_out/roll/pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/roll/pytest.txt | tail -2
