#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r04_check3
mkdir -p $OUT
cd $ROOT
tools/h2d_rate 2>&1 | tee $OUT/h2d_rate.txt
export RPF_PARITY_RECORD=$OUT/fullsize_errors.json
rm -f $RPF_PARITY_RECORD
timeout 1200 python -m pytest tests/test_gpu_heldout.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "held or float32 or thin or split_mixed or large_non_power" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-300
