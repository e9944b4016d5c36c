#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03c4
mkdir -p $OUT
cd $ROOT
export RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so
for pb in 0 8 16 32 64; do
  RPF_TUNE_C4_PIPE=$pb timeout 300 python bench.py --workload C4 --steps 30 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/c4_pipe$pb.json 2> $OUT/c4_pipe$pb.err
done
python - <<'PY'
import json
for pb in (0, 8, 16, 32, 64):
    try:
        d=json.load(open("/root/repo/gpurun_out/r03c4/c4_pipe%d.json"%pb)); r=d["roofline"]
        print("pipe batch %3d: value %.4g ms_per_step %.4f kernel_ms %.4f frac %.3f"%(pb,d["value"],d["ms_per_step"],r["kernel_ms"],r["frac"]))
    except Exception as e: print(pb, "failed", e)
PY
timeout 120 rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0?_(RD|WR)REQ|DRAM|MALL|TCC_HIT|TCC_MISS|TCC_REQ\b" | cut -c1-160 | sort -u | head -40 > $OUT/counters.txt; cat $OUT/counters.txt
