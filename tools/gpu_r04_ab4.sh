#!/bin/bash
# C4 in bench.py's harness: the shipped fused kernel against `make fexp`'s variants of it (librpf_engine_k<n>.so) and the
# two-kernel path, same box.  k3 computes garbage (no raw rows): timing only.
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload C4 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e9,1), 'Gsample/s', round(d['ms_per_step'],4), 'ms, kernel', round(d['roofline']['kernel_ms'],4))"; }
L=$GRAFT_REPO_ROOT/rtl-power-fftw_amd
for rep in 1 2; do
echo "fused (shipped): $(run)"
for k in ${VARIANTS:-3 5 6}; do echo "fused, knob 1 = $k: $(RPF_ENGINE_LIB=$L/librpf_engine_k$k.so run)"; done
echo "two-kernel     : $(run --engine-flags 8)"
done
