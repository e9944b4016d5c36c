#!/bin/bash
# Measurement knobs of the fused four-step kernel's profile build (poll flavour, who polls) by buffer count
cd $GRAFT_REPO_ROOT
export RPF_ENGINE_LIB=$GRAFT_REPO_ROOT/rtl-power-fftw_amd/librpf_engine_fprof.so
for m in 1 2; do for k in "0,0,0,0" "3,0,0,0" "1,0,0,0" "0,1,0,0" "3,1,0,0"; do
  echo "RPF_FUSED_MODE=$m knobs=$k: $(RPF_FUSED_MODE=$m RPF_FUSED_KNOBS=$k timeout 120 python tools/gpu_fused_profile.py 262144 1000 2>&1 | grep -E "^fused|total|wait" | tr '\n' ' ' | cut -c1-330)"
done; done
