#!/bin/bash
# C4 A/B on ONE box: .ab/librpf_head.so (a build of the committed tree, git-ignored) against the working tree:
# four-step parity tests first, then the C4 bench line of each, twice, interleaved.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03c4ab
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fourstep or four_step or 262144 or 65536 or 131072 or C4 or bluestein or large" > $OUT/pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest.txt | tail -3
for i in 1 2; do
  RPF_ENGINE_LIB=$ROOT/.ab/librpf_head.so timeout 300 python bench.py --workload C4 --no-cpu-baseline --no-end-to-end > $OUT/old_$i.json 2>/dev/null
  timeout 300 python bench.py --workload C4 --no-cpu-baseline --no-end-to-end > $OUT/new_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r03c4ab/*_?.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "value %.4g ms_per_step %.5f kernel_ms %.5f"%(d["value"],d["ms_per_step"],r["kernel_ms"]), {k:v for k,v in r.items() if "ms" in k})
    except Exception as e: print(f, "unreadable", e)
PY
