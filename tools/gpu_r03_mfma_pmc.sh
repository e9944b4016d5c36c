#!/bin/bash
# SQ counters of the matrix-pipe first-pass variants beside the VALU kernel (one --pmc pass each, --kernel-trace only)
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03mfma
mkdir -p $OUT
export TMPDIR=/tmp
export RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so
cd $ROOT
for v in 0 60 61 62; do
  python tools/gpu_k1_loop.py 10000 1000 $v > $OUT/plain_$v.txt 2>&1
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_$v -o x -- python $ROOT/tools/gpu_k1_loop.py 10000 200 $v > $OUT/pmc_$v.log 2>&1) || true
done
python - <<'PY'
import csv, glob, collections
out = "/root/repo/gpurun_out/r03mfma"
for v in (0, 60, 61, 62):
    agg = collections.defaultdict(list)
    for f in glob.glob(out + "/pmc_%d/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "fft_accum" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("variant", v, open(out + "/plain_%d.txt" % v).read().strip().splitlines()[-1])
    for k in sorted(agg):
        x = agg[k]; print("   %-28s mean %14.1f  n=%d" % (k, sum(x) / len(x), len(x)))
PY
rm -rf $OUT/pmc_[0-9]*
