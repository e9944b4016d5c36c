#!/bin/bash
# Round 6's GPU jobs, one script, selected by its first argument (gpurun -- 'bash tools/gpu_r06.sh <job>'):
#   census    the WHOLE -m gpu suite with the slow sweep (RPF_RUN_SLOW=1: all seven tone streams), errors recorded, no -x
#   fourstep  the four-step sizes' parity tests, C4's rate and the kernel-only rates on the shipped build, on the
#             float32-pass build (make f32pass) and on the A/B forms of the pass before the last (fswlast / fswdbl =
#             rpf_fourstep.hip compiled with -DRPF_FOURSTEP_WIDE2=0 / =2 and linked with the shipped objects into
#             rtl-power-fftw_amd/librpf_engine_<name>.so), interleaved -- what profiles/r06_fourstep_wide.txt is written from
#   k1        tools/gpu_k1_experiments.py (tuning build): the bounded K1 experiment + board power / clocks
#   c5        C5 with k consecutive scans per launch against a launch per scan: one GPU, --shard-as 2 / 4 / 8, rehearsal
#   final     the driver's suite as the driver runs it (no slow sweep, -x, timed), smoke, the driver's bench line
#   profile   tools/gpu_profile.sh r06 (bench lines, rocprofv3 kernel statistics, PMC passes)
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
cd $ROOT
job=${1:-final}
bench_val() { python3 -c "import json,sys;d=json.load(open('$1'));print('$2', round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step']*1e3,2), 'us/step; kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us;', d['config'].get('scans_per_launch'))" 2>/dev/null || echo "$2: no line ($1)"; }
case $job in
census)
  rm -f $OUT/errors_*.json
  RPF_RUN_SLOW=1 RPF_PARITY_RECORD=$OUT/errors_shipped.json timeout 2400 python -m pytest tests -m gpu -q --durations=60 -p no:cacheprovider > $OUT/census_pytest.log 2>&1; echo "census pytest rc=$?"; tail -25 $OUT/census_pytest.log
  ;;
fourstep)
  # the four-step sizes (65536 ... 262144; C4) on the shipped build (last pass -- behind a radix-4 last pass: last two -- of
  # the row transform in double) and on the float32-pass build (make f32pass): every parity test that touches them, all
  # seven tone streams, errors recorded; C4's rate and the 65536 / 131072 kernel-only rates on both, interleaved
  rm -f $OUT/fs_errors_*.json
  FS='float32 or four_step or fused or c4 or abort or give or held_by or ((thin or picked) and (65536 or 131072 or 262144))'
  for v in ${FS_VARIANTS:-shipped fswlast fswdbl f32pass}; do
    lib=; [ $v != shipped ] && lib=$ROOT/rtl-power-fftw_amd/librpf_engine_$v.so
    RPF_RUN_SLOW=1 RPF_ENGINE_LIB=$lib RPF_PARITY_RECORD=$OUT/fs_errors_$v.json timeout 1500 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fused_abort.py -m gpu -q -p no:cacheprovider -k "$FS" > $OUT/fs_pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -12 $OUT/fs_pytest_$v.log
  done
  for rep in 1 2 3; do
    for v in ${FS_VARIANTS:-shipped fswlast fswdbl f32pass}; do
      lib=; [ $v != shipped ] && lib=$ROOT/rtl-power-fftw_amd/librpf_engine_$v.so
      RPF_ENGINE_LIB=$lib timeout 300 python bench.py --workload C4 --no-cpu-baseline --no-end-to-end > $OUT/c4_${v}_$rep.json 2> $OUT/c4_${v}_$rep.err
      bench_val $OUT/c4_${v}_$rep.json "C4 $v run $rep:"
    done
  done
  for n in 65536 131072 262144; do
    for v in ${FS_VARIANTS:-shipped fswlast fswdbl f32pass}; do
      lib=; [ $v != shipped ] && lib=$ROOT/rtl-power-fftw_amd/librpf_engine_$v.so
      echo "sweep $n $v"; RPF_ENGINE_LIB=$lib SWEEP_K=60 timeout 200 python tools/gpu_sweep.py $n:0 2>&1 | tail -2
    done
  done > $OUT/fourstep_sizes_rate.txt 2>&1
  cat $OUT/fourstep_sizes_rate.txt
  ;;
k1)
  RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tuning.so timeout 900 python tools/gpu_k1_experiments.py 3 > $OUT/k1_experiments.txt 2>&1; echo "k1 rc=$?"; cat $OUT/k1_experiments.txt
  ;;
c5)
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench" > $OUT/c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/c5_pytest.log
  for k in 0 1; do
    timeout 300 python bench.py --workload C5 --scans-per-launch $k --no-cpu-baseline --no-end-to-end > $OUT/c5_one_gpu_k$k.json 2> $OUT/c5_one_gpu_k$k.err; bench_val $OUT/c5_one_gpu_k$k.json "C5 one GPU, --scans-per-launch $k:"
    for n in 2 4 8; do
      timeout 300 python bench.py --workload C5 --shard-as $n --scans-per-launch $k --no-cpu-baseline --no-end-to-end > $OUT/c5_shard_as${n}_k$k.json 2> $OUT/c5_shard_as${n}_k$k.err; bench_val $OUT/c5_shard_as${n}_k$k.json "C5 --shard-as $n, --scans-per-launch $k:"
      timeout 300 python bench.py --workload C5 --shard-as $n --scans-per-launch $k --force-dist --no-cpu-baseline --no-end-to-end > $OUT/c5_shard_as${n}_k${k}_rccl.json 2> $OUT/c5_shard_as${n}_k${k}_rccl.err; bench_val $OUT/c5_shard_as${n}_k${k}_rccl.json "   ... + a one-rank RCCL reduce per 4 scans:"
    done
  done
  for n in 2 8; do
    timeout 600 python bench.py --gpus $n --dist-backend gloo --share-device --steps 50 --warmup 5 > $OUT/c5_${n}rank_gloo.json 2> $OUT/c5_${n}rank_gloo.err; echo "rehearsal $n ranks rc=$?"
    python3 -c "import json;d=json.load(open('$OUT/c5_${n}rank_gloo.json'));print(d['n_gpus'], d['value']/1e9, d['check'], d['rccl'], d['config']['scans_per_launch'])"
  done
  ;;
final)
  rm -f $OUT/fullsize_errors.json
  export RPF_PARITY_RECORD=$OUT/fullsize_errors.json
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -32 $OUT/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
  timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_20.json
  timeout 200 python tools/gpu_stress.py ${2:-120} 77 > $OUT/stress.txt 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress.txt
  ;;
profile)
  bash tools/gpu_profile.sh r06
  ;;
*)
  echo "unknown job $job"; exit 2
  ;;
esac
