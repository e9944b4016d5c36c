"""K1 time as a function of the frame count: slope = steady-state cost per frame, intercept =
per-launch fixed cost (ramp, prologue, partial-spectrum flush, tail).  Back-to-back launches,
HIP events around the whole batch.  Three series:
  single   one acquisition per launch (rpf_device_fused), input cycling through the Infinity Cache
  hbm      the same with every launch reading bytes that were evicted since (ring > 256 MiB)
  scan     H hops of R frames per launch (rpf_device_fused_hops) -- what a hop boundary costs inside
           a launch, and K1 + K3 per scan against H x (K1 + K3)
Usage: python tools/gpu_fixed_cost.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
RMAX = 40000 * 4096 // N
d_in = rpf.synth.noise_tones_iq_torch(2, N * RMAX, dev)
big = [d_in] + [torch.roll(d_in, shifts=2 * N * 37 * i) for i in range(1, 3)]      # 3 x 328 MB
s = torch.cuda.current_stream().cuda_stream


def timed(fn, K=300, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


def fit(rows, fpr, label):
    x = np.array([r for r, _ in rows if r >= fpr], dtype=float); y = np.array([u for r, u in rows if r >= fpr])
    b, a = np.polyfit(x, y, 1)
    print("%s fit: %.2f us fixed + %.3f ns per frame (%.0f ns per full round of %d frames)" % (label, a, b * 1e3, b * 1e3 * fpr, fpr), flush=True)


with rpf.Datastore(rpf.Params(N=N, repeats=RMAX)) as ds:
    ds.device_fused(d_in.data_ptr(), 2 * N * RMAX, RMAX, s)      # (launch_info reports the last launch's grid)
    fpr = ds.launch_info()["grid"] * ds.launch_info()["frames_per_wg"]      # frames per full round
    print("N=%d, %d frames per full round" % (N, fpr))
    sizes = [fpr // 2, fpr, 2 * fpr, 4 * fpr, 8 * fpr, 10000 * 4096 // N, 16 * fpr, 20000 * 4096 // N, 40000 * 4096 // N]
    rows = []
    for R in sizes:
        us = timed(lambda i: ds.device_fused(d_in.data_ptr() + (i % 4) * 2 * N * 16, 2 * N * R, R, s))
        rows.append((R, us))
        print("single R=%6d (%.2f rounds): %8.2f us per launch" % (R, R / fpr, us), flush=True)
    fit(rows, fpr, "single (input in the Infinity Cache)")
    rows = []
    for R in sizes:
        span = 2 * N * R
        per = max(1, (2 * N * RMAX - 2 * N * 64) // max(span, 1))     # distinct windows per 328 MB buffer
        def hbm(i):
            b = big[i % 3]
            ds.device_fused(b.data_ptr() + ((i // 3) % per) * span, span, R, s)
        us = timed(hbm)
        rows.append((R, us))
        print("hbm    R=%6d (%.2f rounds): %8.2f us per launch" % (R, R / fpr, us), flush=True)
    fit(rows, fpr, "hbm (input evicted between uses)")
    # scans: H hops of R frames in one launch
    out = torch.empty((16, N), dtype=torch.float64, device=dev)
    for H, R in ((8, 5000 * 4096 // N), (8, 625 * 4096 // N), (16, 2500 * 4096 // N), (1, 40000 * 4096 // N)):
        ptrs = [big[h % 3].data_ptr() + (h // 3) * 2 * N * R for h in range(H)]
        nb = [2 * N * R] * H
        k1 = timed(lambda i: ds.device_fused_hops(ptrs, nb, [R] * H, s), K=100)
        both = timed(lambda i: ds.accumulate_device_hops(ptrs, nb, [R] * H, out.data_ptr(), s), K=100)
        def per_hop(i):
            for h in range(H):
                ds.accumulate_device(ptrs[h], nb[h], R, out[h].data_ptr(), s)
        old = timed(per_hop, K=100)
        print("scan   H=%2d x R=%6d: K1 alone %8.2f us, K1+K3 %8.2f us per scan (%.3f Tsample/s); hop by hop %8.2f us (%.3f Tsample/s)"
              % (H, R, k1, both, H * R * N / both / 1e6, old, H * R * N / old / 1e6), flush=True)
