cd $GRAFT_REPO_ROOT
timeout 120 rocgdb -batch -ex "set pagination off" -ex "set amdgpu precise-memory on" -ex run -ex "x/4i \$pc-16" -ex "info registers s0 s1 s2 s3 s4 s5 s6 s7 s8 s9 s10 s11" --args tools/l2_residency_bench 10 6 2 2>&1 | grep -v "^\[New\|^\[Thread\|warning" | head -40 | cut -c1-250
