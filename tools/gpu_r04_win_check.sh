#!/bin/bash
# Windowed runs of every split-form size of round 3's windowed sweep on today's tree (regressions against
# profiles/r03_windowed_sizes.txt?), 32768 windowed on both four-step forms, and the tests that name 32768.
cd $GRAFT_REPO_ROOT
SWEEP_ONLYWIN=1 SWEEP_K=60 timeout 400 python tools/gpu_sweep.py 11000:0 12500:0 12800:0 13000:0 14400:0 15360:0 16000:0 16384:0 20000:0 24000:0 25000:0 30000:0 32000:0 32768:0 36000:0 40000:0 45000:0 48000:0 50000:0 54000:0 55000:0 56000:0 60000:0 64000:0 66000:0 68000:0 72000:0 75000:0 76000:0 80000:0 81000:0 81920:0 88000:0 90000:0 92000:0 96000:0 98304:0 100000:0 104000:0 105000:0 108000:0  > gpurun_out/win_now.txt 2>&1
grep -c Gsample gpurun_out/win_now.txt
grep "N=32768" gpurun_out/win_now.txt | cut -c1-140
SWEEP_ONLYWIN=1 SWEEP_K=60 SWEEP_FLAGS=8 timeout 100 python tools/gpu_sweep.py 32768:0 2>&1 | grep "N=32768" | cut -c1-140
timeout 500 python -m pytest tests -m gpu -x -q -k "32768" 2>&1 | tail -2
