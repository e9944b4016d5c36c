#!/bin/bash
# A/B on ONE box: another tree (e.g. `git archive <rev> | tar -x -C .ab_r02` + make, git-ignored) against the working
# tree, and the scan kernel on one-hop scans (tuning build) beside both: K1 fixed cost + the driver-style bench line.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03ab
mkdir -p $OUT
for i in 1 2; do
  (cd $ROOT/.ab_r02 && timeout 200 python tools/gpu_fixed_cost.py) > $OUT/old_$i.txt 2>&1
  (cd $ROOT && timeout 200 python tools/gpu_fixed_cost.py) > $OUT/new_$i.txt 2>&1
  (cd $ROOT && RPF_TUNE_SCAN_KERNEL=1 RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so timeout 200 python tools/gpu_fixed_cost.py) > $OUT/contig_$i.txt 2>&1
  (cd $ROOT/.ab_r02 && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end) > $OUT/old_bench_$i.json 2>/dev/null
  (cd $ROOT && timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end) > $OUT/new_bench_$i.json 2>/dev/null
done
grep -h "fit\|R= 10000\|scan" $OUT/old_*.txt $OUT/new_*.txt $OUT/contig_*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r03ab/*_bench_*.json")):
    d=json.load(open(f)); r=d["roofline"]
    print(f.split("/")[-1], "value %.4g ms_per_step %.5f kernel_ms %.5f"%(d["value"],d["ms_per_step"],r["kernel_ms"]))
PY
