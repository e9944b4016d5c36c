#!/bin/bash
# On the GPU box: round 3's new split-form sizes -- the shipped forms (plain, windowed), the windowed candidates
# (winsearch) and the one-buffer twins (gen_roll_candidates, ROLL_BASE=41) of the tuning build, then the parity tests.
#   tools/gpu_new_sizes.sh "<sizes>" "<sizes with roll twins>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/new
export SWEEP_K=${SWEEP_K:-100}
SHIPPED=$(for n in $1; do echo -n "$n:0 "; done)
timeout 600 python tools/gpu_sweep.py $SHIPPED > gpurun_out/new/shipped.txt 2>&1
export RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so
SWEEP_ONLYWIN=1 timeout 900 python tools/gpu_sweep.py $(python tools/gen_mixed_plans.py wincases $1) > gpurun_out/new/win_search.txt 2>&1
ROLL_BASE=41 ROLL_SIZES="$2" timeout 600 python tools/gpu_sweep.py $(ROLL_BASE=41 ROLL_SIZES="$2" python tools/gen_roll_candidates.py cases) > gpurun_out/new/roll_search.txt 2>&1
unset RPF_ENGINE_LIB
grep -c Gsample gpurun_out/new/*.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mixed_radix or thin" > gpurun_out/new/pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/new/pytest.txt | tail -2
