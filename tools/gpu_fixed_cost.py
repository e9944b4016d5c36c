"""K1 time as a function of the frame count: slope = steady-state cost per frame, intercept =
per-launch fixed cost (ramp, prologue, partial-spectrum flush, tail).  Back-to-back launches,
HIP events around the whole batch.  Usage: python tools/gpu_fixed_cost.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
RMAX = 40000 * 4096 // N
d_in = rpf.synth.noise_tones_iq_torch(2, N * RMAX, dev)
s = torch.cuda.current_stream().cuda_stream
rows = []
with rpf.Datastore(rpf.Params(N=N, repeats=RMAX)) as ds:
    fpr = ds.launch_info()["grid"] * ds.launch_info()["frames_per_wg"]      # frames per full round
    for R in [fpr // 2, fpr, 2 * fpr, 4 * fpr, 8 * fpr, 10000 * 4096 // N, 16 * fpr, 20000 * 4096 // N, 40000 * 4096 // N]:
        for _ in range(20):
            ds.device_fused(d_in.data_ptr(), 2 * N * R, R, s)
        torch.cuda.synchronize()
        K = 300
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            off = (i % 4) * 2 * N * 16
            ds.device_fused(d_in.data_ptr() + off, 2 * N * R, R, s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        rows.append((R, us))
        print("R=%6d (%.2f rounds): %8.2f us per launch" % (R, R / fpr, us), flush=True)
x = np.array([r for r, _ in rows if r >= fpr], dtype=float); y = np.array([u for r, u in rows if r >= fpr])
b, a = np.polyfit(x, y, 1)
print("fit: %.2f us fixed + %.3f ns per frame (%.0f ns per full round of %d frames)" % (a, b * 1e3, b * 1e3 * fpr, fpr))
