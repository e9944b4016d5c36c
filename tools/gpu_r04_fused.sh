#!/bin/bash
# Round 4: the fused four-step kernel (v3) on the GPU box: parity against the two-kernel path, where a round's time goes,
# and the rate at config C4's size.  Output: gpurun_out/r04_fused/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_fused
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_four_step" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_fprof.so timeout 200 python tools/gpu_fused_profile.py 262144 1000 > $OUT/profile_262144.txt 2>&1; echo "profile rc=$?"; cat $OUT/profile_262144.txt
timeout 200 python tools/gpu_fused_profile.py 262144 1000 > $OUT/rate_262144.txt 2>&1; cat $OUT/rate_262144.txt
for n in 65536 131072; do timeout 200 python tools/gpu_fused_profile.py $n 2000 > $OUT/rate_$n.txt 2>&1; cat $OUT/rate_$n.txt; done
