cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_four_step and 262144" 2>&1 | grep -B2 -A6 "AssertionError" | head -40
