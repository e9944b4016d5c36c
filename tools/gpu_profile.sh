#!/bin/bash
# Runs on the GPU box (via gpurun): bench.py, then the same command under
# rocprofv3 --kernel-trace --stats, then PMC passes (each in its own run, as the
# HBM/rocprofv3 section of MI355X_MICROARCH.md prescribes).  Output under
# gpurun_out/$1/ ; summaries are copied to profiles/ by tools/summarize_profile.py.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c2 -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o c2 -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o c2 -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o c2 -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o c2 -- $BENCH > $OUT/pmc_lds.log 2>&1
# config C3 (Hann window): kernel trace + HBM byte counters of the windowed kernel
BENCH3="$BENCH --workload C3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o c3 -- $BENCH3 > $OUT/c3_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c3_pmc_fetch -o c3 -- $BENCH3 > $OUT/c3_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c3_pmc_write -o c3 -- $BENCH3 > $OUT/c3_pmc_write.log 2>&1
(cd $GRAFT_REPO_ROOT && python bench.py --workload C3 --no-cpu-baseline > $OUT/c3_bench.json 2> $OUT/c3_bench.err)
# config C4 (N = 262144 x 1000, four-step kernels)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -o c4 -- python $GRAFT_REPO_ROOT/tools/gpu_c4.py 10 > $OUT/c4.json 2> $OUT/c4.err
[ -x $GRAFT_REPO_ROOT/tools/hbm_read_bench ] && timeout 120 $GRAFT_REPO_ROOT/tools/hbm_read_bench > $OUT/hbm_read.txt 2>&1
ls -R $OUT | head -40
