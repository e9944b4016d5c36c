#!/bin/bash
# Runs on the GPU box (via gpurun): bench.py lines, then the same commands under
# rocprofv3 --kernel-trace --stats, then PMC passes (each in its own run, as the
# HBM/rocprofv3 section of MI355X_MICROARCH.md prescribes).  Output under
# gpurun_out/$1/ ; summaries are copied to profiles/ by tools/summarize_profile.py.
set -u
TAG=${1:-r03}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
date -u +%Y-%m-%dT%H:%M:%SZ > $OUT/captured.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_20steps.json 2>> $OUT/bench.err
python bench.py --workload C3 --no-cpu-baseline --no-end-to-end > $OUT/c3_bench.json 2> $OUT/c3_bench.err
python bench.py --workload C4 --no-end-to-end > $OUT/c4_bench.json 2> $OUT/c4_bench.err
python bench.py --workload C5 --no-cpu-baseline --no-end-to-end > $OUT/c5_bench.json 2> $OUT/c5_bench.err
python bench.py --workload C5 --force-dist --no-cpu-baseline > $OUT/c5_rccl_bench.json 2>> $OUT/c5_bench.err
python bench.py --workload C5 --shard-as 8 --no-cpu-baseline --no-end-to-end > $OUT/c5_shard8_bench.json 2>> $OUT/c5_bench.err
cut -c1-200 $OUT/bench.json $OUT/c4_bench.json $OUT/c5_bench.json
cd /tmp && export TMPDIR=/tmp
B="--steps 400 --warmup 40 --no-cpu-baseline --no-end-to-end"
prof() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o $name -- python $ROOT/bench.py "$@" > $OUT/${name}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${name}_pmc_$c -o $name -- python $ROOT/bench.py "$@" > $OUT/${name}_pmc_$c.log 2>&1
  done
}
prof c2 $B
prof c3 $B --workload C3
prof c4 --workload C4 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end
prof c5 --workload C5 --steps 100 --warmup 10 --no-cpu-baseline --no-end-to-end
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/c2_pmc_sq -o c2 -- python $ROOT/bench.py $B > $OUT/c2_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c2_pmc_lds -o c2 -- python $ROOT/bench.py $B > $OUT/c2_pmc_lds.log 2>&1
cd $ROOT
[ -x tools/hbm_read_bench ] && timeout 120 tools/hbm_read_bench > $OUT/hbm_read.txt 2>&1
[ -x tools/lds_valu_bench ] && timeout 120 tools/lds_valu_bench > $OUT/lds_valu.txt 2>&1
timeout 200 python tools/gpu_fixed_cost.py > $OUT/k1_fixed_cost.txt 2>&1
[ -x tools/mfma_valu_bench ] && timeout 120 tools/mfma_valu_bench > $OUT/mfma_valu.txt 2>&1
timeout 600 python tools/gpu_sweep.py 64:0 128:0 256:0 512:0 1024:0 2048:0 4096:0 8192:0 16384:0 32768:0 65536:0 131072:0 262144:0 500:0 1000:0 3000:0 4094:0 5000:0 20000:0 100000:0 131070:0 524288:0 > $OUT/sizes.txt 2>&1
# condense here: the raw rocprofv3 output is far beyond what gpurun copies back
python tools/summarize_profile.py $TAG > $OUT/summarize.log 2>&1
tail -3 $OUT/summarize.log
for d in $OUT/*_trace $OUT/*_pmc_* ; do [ -d "$d" ] && rm -rf "$d"; done
du -sh $OUT; ls $OUT/summary
