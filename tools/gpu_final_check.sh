#!/bin/bash
# Round-end verification on the GPU box: the -m gpu suite, smoke(), the driver's bench command,
# a rocprofv3 kernel-trace of the mixed-radix sizes, then the randomised stress for the rest of the budget.
# QUICK=1: without the kernel trace and the size sweep (a change that touches one kernel only).
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
rm -f $OUT/fullsize_errors.json
export RPF_PARITY_RECORD=$OUT/fullsize_errors.json      # the parity tests' measured errors -> profiles/rNN_fullsize_errors.json
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_20.json
if [ -z "${QUICK:-}" ]; then
cd /tmp && export TMPDIR=/tmp
SWEEP_NOWIN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/km_trace -o km -- python $ROOT/tools/gpu_sweep.py 500:0 1000:0 2000:0 5000:0 7000:0 10000:0 15000:0 16384:0 20000:0 32768:0 50000:0 > $OUT/km_trace.log 2>&1
cp $(find $OUT/km_trace -name 'km_kernel_stats.csv' | head -1) $OUT/km_kernel_stats.csv 2>/dev/null
rm -rf $OUT/km_trace
cd $ROOT
fi
timeout 200 python tools/gpu_stress.py ${1:-120} 77 > $OUT/stress.txt 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress.txt
[ -z "${QUICK:-}" ] && timeout 300 python tools/gpu_sweep.py 64:0 128:0 256:0 512:0 1024:0 2048:0 4096:0 8192:0 16384:0 32768:0 65536:0 131072:0 262144:0 100:0 500:0 1000:0 1536:0 2000:0 3000:0 4000:0 5000:0 6000:0 7000:0 10000:0 12000:0 15000:0 20000:0 30000:0 40000:0 50000:0 60000:0 1458:0 4094:0 7002:0 14000:0 100000:0 131070:0 524288:0 > gpurun_out/sizes.txt 2>&1; grep -c Gsample gpurun_out/sizes.txt
# C4 on both four-step paths, same box, bench.py's own step counts
for f in 0 8; do timeout 300 python bench.py --workload C4 --engine-flags $f --no-cpu-baseline --no-end-to-end > $OUT/bench_c4_flags$f.json 2>/dev/null; python3 -c "import json;d=json.load(open('$OUT/bench_c4_flags$f.json'));print('C4 engine-flags $f:', d['value']/1e9, 'Gsample/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])"; done
