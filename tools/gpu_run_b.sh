#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export RPF_ENGINE_LIB=$GRAFT_REPO_ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_NOWIN=1
timeout 600 python tools/gpu_sweep.py 4096:0 4096:9 4096:48 4096:43 4096:44 4096:45 4096:49 4096:10 4096:46 4096:47 4096:9 4096:44 > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
cut -c1-150 $OUT/sweep.log
