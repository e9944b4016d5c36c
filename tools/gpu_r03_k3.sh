#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03k3
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
(cd $ROOT/.ab_r02 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end) > $OUT/old_$rep.json 2>/dev/null
for k in 0 1 2 3 4 5 6; do
  RPF_TUNE_K3=$k RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/k3_${k}_$rep.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r03k3/*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("/")[-1], "ms_per_step %.5f kernel_ms %.5f  rest %.2f us"%(d["ms_per_step"],r["kernel_ms"],1e3*(d["ms_per_step"]-r["kernel_ms"])))
PY
