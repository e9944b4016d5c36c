"""K1 alone, back to back, on one replay buffer (for rocprofv3 counter passes and A/B runs of two
library builds).  Usage: python tools/gpu_k1_loop.py [R] [launches] [variant]
(variants other than 0: RPF_ENGINE_LIB=.../librpf_engine_tuning.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf
N = 4096
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
VID = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
d_in = rpf.synth.noise_tones_iq_torch(2, N * R, dev)
s = torch.cuda.current_stream().cuda_stream
with rpf.Datastore(rpf.Params(N=N, repeats=R), flags=VID << 8) as ds:
    for _ in range(3000):          # clocks up
        ds.device_fused(d_in.data_ptr(), 2 * N * R, R, s)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        ds.device_fused(d_in.data_ptr(), 2 * N * R, R, s)
    e1.record(); torch.cuda.synchronize()
    print("R=%d variant %d: %.2f us per launch" % (R, VID, e0.elapsed_time(e1) / K * 1e3))
