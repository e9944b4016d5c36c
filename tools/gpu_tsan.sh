#!/bin/bash
# ThreadSanitizer pass over the host concurrency, on the GPU box.  Output: gpurun_out/tsan/
#   1. tools/tsan_queue_driver (C++: queue protocol with straddling frames, unget / early finish, large-buffer
#      rpf_accumulate, two engines from two threads)
#   2. the CLI's several-engines test (tests/test_host.py::test_cli_scan_spread_over_several_engines) with rpf_power_tsan
#   3. the Python queue tests with the sanitised engine library preloaded into an uninstrumented python
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/tsan
mkdir -p $OUT
cd $ROOT
make -C rtl-power-fftw_amd/host tsan > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(make -s -C rtl-power-fftw_amd/csrc tsan-runtime)
$CXX -std=c++17 -O1 -g -fsanitize=thread -shared-libsan -o tools/tsan_queue_driver tools/tsan_queue_driver.cpp \
   -Lrtl-power-fftw_amd -lrpf_engine_tsan -Wl,-rpath,$ROOT/rtl-power-fftw_amd -Wl,-rpath,$(dirname $RT) -Wl,-rpath,/opt/rocm/lib -lpthread >> $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
# the HIP runtime and its helper threads are not instrumented: reports whose every frame is inside them are not ours
cat > $OUT/suppressions.txt <<SUP
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUP
export TSAN_OPTIONS="halt_on_error=0 exitcode=66 second_deadlock_stack=1 suppressions=$OUT/suppressions.txt"
echo "== 1. tsan_queue_driver"; timeout 600 tools/tsan_queue_driver > $OUT/driver.log 2>&1; echo "rc=$?"; tail -3 $OUT/driver.log
echo "== 2. CLI, several engines"; RPF_POWER_CLI=$ROOT/rtl-power-fftw_amd/host/rpf_power_tsan timeout 900 python -m pytest tests/test_host.py -m gpu -x -q -k "several_engines or cli_file_replay" > $OUT/cli.log 2>&1; echo "rc=$?"; tail -3 $OUT/cli.log
echo "== 3. python queue tests, preloaded runtime"; LD_PRELOAD=$RT RPF_NO_TORCH=1 RPF_ENGINE_LIB=$ROOT/rtl-power-fftw_amd/librpf_engine_tsan.so timeout 900 setarch -R python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_abort.py -m gpu -x -q -k "buffer_queue_path or unget_and_early or protocol_errors or neighbouring or registered or falls_back or one_launch or buffer_protocol" > $OUT/pyqueue.log 2>&1; echo "rc=$?"; tail -3 $OUT/pyqueue.log
echo "== reports"; grep -c "WARNING: ThreadSanitizer" $OUT/driver.log $OUT/cli.log $OUT/pyqueue.log
grep -A12 "WARNING: ThreadSanitizer" $OUT/driver.log $OUT/cli.log $OUT/pyqueue.log | head -120
