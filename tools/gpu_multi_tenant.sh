cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 300 python bench.py --workload C2 --gpus 2 --dist-backend gloo --share-device --steps 40 --warmup 5 > gpurun_out/r05/c2_2rank_gloo.json 2> gpurun_out/r05/c2_2rank_gloo.err; echo "C2 weak 2 ranks rc=$?"
python3 -c "import json;d=json.load(open('gpurun_out/r05/c2_2rank_gloo.json'));print(d['n_gpus'], d['scaling'], d['value']/1e9, d['config']['reduce'], len(d['per_rank']))"
timeout 600 python bench.py --workload C4 --gpus 2 --dist-backend gloo --share-device --steps 6 --warmup 1 > gpurun_out/r05/c4_2rank_gloo.json 2> gpurun_out/r05/c4_2rank_gloo.err; echo "C4 two tenants rc=$?"
python3 -c "import json;d=json.load(open('gpurun_out/r05/c4_2rank_gloo.json'));print(d['n_gpus'], d['value']/1e9, d['ms_per_step'], d['roofline']['fused_four_step'], d['roofline']['kernel'][:50])"
tail -5 gpurun_out/r05/c4_2rank_gloo.err
# two CLI-like processes on one device: both run a queue-fed C4-size acquisition at once
cat > /tmp/tenant.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import rtl_power_fftw_amd as rpf
N, R = 262144, 400
stream = rpf.synth.noise_tones_iq(4, N * R)
with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
    for i in range(6):
        pwr, done = ds.accumulate(stream, R)
        assert done == R and np.all(np.isfinite(pwr))
        print(sys.argv[1], i, ds.fused_status(), float(pwr.sum()), flush=True)
PY
python /tmp/tenant.py A > gpurun_out/r05/tenant_a.txt 2>&1 &
python /tmp/tenant.py B > gpurun_out/r05/tenant_b.txt 2>&1 &
wait
cat gpurun_out/r05/tenant_a.txt gpurun_out/r05/tenant_b.txt | grep -v amdgpu.ids
