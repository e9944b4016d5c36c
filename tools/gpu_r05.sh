#!/bin/bash
# Round 5's GPU jobs, one script, selected by its first argument (gpurun -- 'bash tools/gpu_r05.sh <job>'):
#   ranks     bench.py's self-launched multi-rank runs rehearsed on one GPU (2/4/8 ranks, gloo, shared device),
#             the per-rank step of a 2/4/8-rank C5 job (--shard-as), and the launcher tests
#   fused     the fused four-step kernel's abort / fall-back tests and C4 bench on both paths
#   stream    the buffer-queue rates (bench.py's end_to_end leg + tools/queue_rate)
#   final     tools/gpu_final_check.sh
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
job=${1:-ranks}
case $job in
ranks)
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench" > $OUT/ranks_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/ranks_pytest.log
  for n in 2 4 8; do
    timeout 600 python bench.py --gpus $n --dist-backend gloo --share-device --steps 50 --warmup 5 > $OUT/c5_${n}rank_gloo.json 2> $OUT/c5_${n}rank_gloo.err; echo "rehearsal $n ranks rc=$?"
    python3 -c "import json;d=json.load(open('$OUT/c5_${n}rank_gloo.json'));print(d['n_gpus'], d['value']/1e9, d['check'], [ (p['rank'],p['frames_per_step'],round(p['kernel_ms_median'],4)) for p in d['per_rank']])"
  done
  for n in 1 2 4 8; do
    timeout 300 python bench.py --workload C5 --shard-as $n --no-cpu-baseline --no-end-to-end > $OUT/c5_shard_as$n.json 2> $OUT/c5_shard_as$n.err; echo "shard-as $n rc=$?"
    python3 -c "import json;d=json.load(open('$OUT/c5_shard_as$n.json'));print('shard-as $n: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
  for n in 2 4 8; do
    timeout 300 python bench.py --workload C5 --shard-as $n --force-dist --no-cpu-baseline --no-end-to-end > $OUT/c5_shard_as${n}_rccl.json 2> $OUT/c5_shard_as${n}_rccl.err; echo "shard-as $n rccl rc=$?"
    python3 -c "import json;d=json.load(open('$OUT/c5_shard_as${n}_rccl.json'));print('shard-as $n + 1-rank RCCL reduce: ms_per_step', d['ms_per_step'])"
  done
  ;;
fused)
  timeout 900 python -m pytest tests/test_gpu_fused_abort.py -m gpu -x -q -rs > $OUT/fused_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/fused_pytest.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_step or fused" > $OUT/fused_parity.log 2>&1; echo "parity rc=$?"; tail -5 $OUT/fused_parity.log
  for f in 0 8; do timeout 300 python bench.py --workload C4 --engine-flags $f --no-cpu-baseline --no-end-to-end > $OUT/bench_c4_flags$f.json 2>/dev/null; python3 -c "import json;d=json.load(open('$OUT/bench_c4_flags$f.json'));print('C4 engine-flags $f:', d['value']/1e9, 'Gsample/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'], d['roofline']['kernel'][:40])"; done
  ;;
wide)
  # the split / paired forms with the wide (double) last pass against their float32 last pass: rate of every size, then
  # the held-out and tone-stream parity of the shipped (wide) build
  SPLIT="${SPLIT:-21000 32000 34000 35000 40000 42000 44000 45000 46000 48000 49000 50000 51000 52000 54000 55000 56000 57000 60000 63000 64000 65000 66000 68000 69000 70000 72000 75000 76000 77000 78000 80000 81000 81920 88000 90000 92000 96000 98304 100000 104000 105000 108000}"
  CASES=$(for n in $SPLIT; do echo -n "$n:0 "; done)
  SWEEP_K=100 timeout 900 python tools/gpu_sweep.py $CASES > $OUT/sweep_wide.txt 2>&1; echo "sweep wide rc=$?"
  RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_nowide.so SWEEP_K=100 timeout 900 python tools/gpu_sweep.py $CASES > $OUT/sweep_nowide.txt 2>&1; echo "sweep nowide rc=$?"
  rm -f $OUT/heldout_wide.json
  RPF_PARITY_RECORD=$OUT/heldout_wide.json timeout 2400 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_parity.py -m gpu -q -k "held or picked or float32 or thin or split or mixed or registered or neighbouring or queue or unget or protocol" > $OUT/heldout_wide.log 2>&1; echo "heldout rc=$?"; tail -15 $OUT/heldout_wide.log
  ;;
halfframe)
  # the fused four-step kernel's half-frame form (make nbuf3) against the shipped two-buffer form: parity of every four-step
  # size and the give-up path on the variant, C4 rate A/B on bench.py (interleaved), PMC traffic of both
  NB3=$ROOT/rtl-power-fftw_amd/librpf_engine_nbuf3.so
  RPF_ENGINE_LIB=$NB3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_abort.py -m gpu -x -q -k "four_step or fused or abort or give or held_by" > $OUT/half_pytest.log 2>&1; echo "pytest(nbuf3) rc=$?"; tail -4 $OUT/half_pytest.log
  for rep in 1 2 3; do
    for v in shipped nbuf3; do
      lib=; [ $v = nbuf3 ] && lib=$NB3
      RPF_ENGINE_LIB=$lib timeout 300 python bench.py --workload C4 --no-cpu-baseline --no-end-to-end > $OUT/half_bench_${v}_$rep.json 2>/dev/null
      python3 -c "import json;d=json.load(open('$OUT/half_bench_${v}_$rep.json'));print('C4 $v run $rep:', round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms; kernel', round(d['roofline']['kernel_ms'],4))"
    done
  done
  for n in 65536 131072; do
    for v in shipped nbuf3; do
      lib=; [ $v = nbuf3 ] && lib=$NB3
      RPF_ENGINE_LIB=$lib SWEEP_NOWIN=1 SWEEP_K=60 timeout 200 python tools/gpu_sweep.py $n:0 2>&1 | grep Gsample | sed "s/^/$v /"
    done
  done
  cd /tmp && export TMPDIR=/tmp
  for v in shipped nbuf3; do
    lib=; [ $v = nbuf3 ] && lib=$NB3
    for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
      RPF_ENGINE_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/half_pmc_$v/$c -o c4 -- python $ROOT/tools/gpu_fused_profile.py 262144 512 > $OUT/half_pmc_${v}_$c.log 2>&1
    done
    python3 - <<PY
import csv, collections, glob
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    acc = []
    for f in glob.glob("$OUT/half_pmc_$v/%s/**/c4_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fused_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc.append(float(r["Counter_Value"]))
    acc = sorted(acc)[len(acc)//2:]
    if acc: print("$v", c, "fused launches", len(acc), "median per 512-frame launch %.0f" % acc[len(acc)//2])
PY
    rm -rf $OUT/half_pmc_$v
  done
  cd $ROOT
  ;;
fswide)
  # the four-step kernels with the last pass of their row transform in double (make fswide) against the shipped float32
  # pass: errors on the tone streams at 131072 / 262144 and on C4's own stream, C4 rate on bench.py (interleaved)
  FSW=$ROOT/rtl-power-fftw_amd/librpf_engine_fswide.so
  for v in shipped fswide; do
    lib=; [ $v = fswide ] && lib=$FSW
    RPF_ENGINE_LIB=$lib RPF_PARITY_RECORD=$OUT/fswide_errors_$v.json timeout 900 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "float32 or (thin and (131072 or 262144)) or four_step or fused_four or c4" > $OUT/fswide_pytest_$v.log 2>&1; echo "pytest($v) rc=$?"; tail -3 $OUT/fswide_pytest_$v.log
  done
  for rep in 1 2 3; do
    for v in shipped fswide; do
      lib=; [ $v = fswide ] && lib=$FSW
      RPF_ENGINE_LIB=$lib timeout 300 python bench.py --workload C4 --no-cpu-baseline --no-end-to-end > $OUT/fswide_bench_${v}_$rep.json 2>/dev/null
      python3 -c "import json;d=json.load(open('$OUT/fswide_bench_${v}_$rep.json'));print('C4 $v run $rep:', round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms; kernel', round(d['roofline']['kernel_ms'],4))"
    done
  done
  ;;
stream)
  tools/h2d_rate > $OUT/h2d_rate.txt 2>&1; echo "h2d_rate rc=$?"; cat $OUT/h2d_rate.txt
  LD_LIBRARY_PATH=$ROOT/rtl-power-fftw_amd tools/queue_rate > $OUT/queue_rate.txt 2>&1; echo "queue_rate rc=$?"; cat $OUT/queue_rate.txt
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench rc=$?"
  python3 -c "import json;d=json.load(open('$OUT/bench_20.json'));print(d['value']/1e9, d['ms_per_step']); [print(c) for c in d['end_to_end']['cases']]"
  timeout 900 python -m pytest tests/test_gpu_fused_abort.py tests/test_gpu_parity.py -m gpu -x -q -k "abort or give or held_by or fused or registered or neighbouring or queue or unget or protocol" > $OUT/stream_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/stream_pytest.log
  ;;
tsan)
  bash tools/gpu_tsan.sh
  ;;
final)
  shift; bash tools/gpu_final_check.sh "$@"
  ;;
*) echo "unknown job $job"; exit 1;;
esac
