#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03mfma
mkdir -p $OUT
cd $ROOT
RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so SWEEP_NOWIN=1 timeout 300 python tools/gpu_sweep.py 4096:0 4096:60 4096:61 4096:62 4096:0 4096:62 > $OUT/sweep.txt 2>&1; cat $OUT/sweep.txt
