#!/bin/bash
# Fused four-step kernel against the two-kernel path, all five four-step sizes, windowed and not (81.92 MB per launch,
# kernel-only HIP events; parity of 64 frames against the f64 oracle in the same lines).  RPF_FUSED_MODE: measurement
# only -- 0/1: two buffers of Y per team (nt hints on / off), 2/3: one buffer (off / on).
cd $GRAFT_REPO_ROOT
SIZES="16384:0 32768:0 65536:0 131072:0 262144:0"
echo "== two-kernel path"; SWEEP_K=50 SWEEP_FLAGS=12 timeout 300 python tools/gpu_sweep.py $SIZES 2>&1 | grep -v amdgpu.ids
for m in 1 2; do echo "== fused, RPF_FUSED_MODE=$m"; RPF_FUSED_MODE=$m SWEEP_K=50 SWEEP_FLAGS=2 timeout 300 python tools/gpu_sweep.py $SIZES 2>&1 | grep -v amdgpu.ids; done
