#!/bin/bash
# On the GPU box: time the windowed twins of the split-form candidates (tuning build made with
# PLAN_MODE=winsearch), beside the shipped kernels' plain and windowed rates for the same sizes.
#   tools/gpu_win_search.sh <seconds> sizes ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SWEEP_K=${SWEEP_K:-120}
SHIPPED=$(for n in ${@:2}; do echo -n "$n:0 "; done)
timeout ${1:-900} python tools/gpu_sweep.py $SHIPPED > gpurun_out/win_shipped.txt 2>&1
export RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_ONLYWIN=1
timeout ${1:-900} python tools/gpu_sweep.py $(python tools/gen_mixed_plans.py wincases ${@:2}) > gpurun_out/win_search.txt 2>&1
grep -c "Gsample" gpurun_out/win_shipped.txt gpurun_out/win_search.txt; grep -v Gsample gpurun_out/win_search.txt | head -5
