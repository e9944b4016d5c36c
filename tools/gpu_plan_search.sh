#!/bin/bash
# On the GPU box: time every candidate plan of the tuning build (tools/gen_mixed_plans.py).
#   tools/gpu_plan_search.sh <seconds> [sizes ...]        MODE=cases2 for a "search2" build
cd $GRAFT_REPO_ROOT
export RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_NOWIN=1
timeout ${1:-900} python tools/gpu_sweep.py $(python tools/gen_mixed_plans.py ${MODE:-cases} ${@:2}) > gpurun_out/plan_search.txt 2>&1
grep -c "Gsample" gpurun_out/plan_search.txt; grep -v Gsample gpurun_out/plan_search.txt | head -5
