"""Round 6's bounded K1 experiment: the C2 step (K1 + K3, nine rotating replay buffers in HBM like
bench.py) on the shipped kernel and on the tuning build's accumulate variants, interleaved on one box; each variant's
spectrum against the shipped kernel's and float64 truth; the board's power and clocks while the shipped kernel runs.
Usage (GPU box): RPF_ENGINE_LIB=.../librpf_engine_tuning.so python tools/gpu_k1_experiments.py [rounds]
  variant 0   shipped: two v_cvt_f64_f32 + two v_fma_f64 per bin and frame
  variant 22  float32 pre-accumulate over 8 frames, one float per bin (two v_fma_f32 per bin and frame; round 3)
  variant 23  PACKED float32 pre-accumulate over 8 frames ((re^2, im^2) apart: one v_pk_fma_f32 per bin and frame)
  variant 24  the same over 16 frames
  variant 31  ablation: no accumulate at all (the ceiling of anything done to the accumulate; wrong results)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import rtl_power_fftw_amd as rpf
from helpers import truth_f64

N, R = 4096, 10000
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
VARIANTS = [0, 22, 23, 24, 31]
dev = torch.device("cuda:0")
base = rpf.synth.noise_tones_iq_torch(2, N * R, dev)
bufs = [base] + [torch.roll(base, shifts=2 * N * 37 * i) for i in range(1, 9)]
s = torch.cuda.current_stream().cuda_stream
d_out = torch.zeros(N, dtype=torch.float64, device=dev)
engines = {v: rpf.Datastore(rpf.Params(N=N, repeats=R), flags=v << 8) for v in VARIANTS}


def steps(ds, k):
    for i in range(k):
        ds.device_fused(bufs[i % 9].data_ptr(), 2 * N * R, R, s)
        ds.device_reduce(d_out.data_ptr(), s)


def timed(ds, k=1500):
    steps(ds, 300)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps(ds, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


# accuracy: 64 frames (little averaging: the pre-accumulate's own rounding shows) and the whole acquisition
host64 = base[: 2 * N * 64].cpu().numpy()
truth64 = truth_f64(N, host64, 64, None)
ref = {}
for v, ds in engines.items():
    out = {}
    for frames in (64, R):
        ds.accumulate_device(base.data_ptr(), 2 * N * frames, frames, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        out[frames] = d_out.cpu().numpy().copy()
    ref[v] = out
print("variant  vs shipped (64 fr)  vs shipped (10000 fr)  vs float64 truth (64 fr)")
for v in VARIANTS:
    if v == 31:
        continue
    print("%7d  %16.2e  %21.2e  %24.2e" % (
        v, np.max(np.abs(ref[v][64] - ref[0][64]) / ref[0][64]), np.max(np.abs(ref[v][R] - ref[0][R]) / ref[0][R]),
        np.max(np.abs(ref[v][64] - truth64) / truth64)))

print("\nC2 step (K1 + K3), us, %d interleaved rounds:" % ROUNDS)
table = {v: [] for v in VARIANTS}
for _ in range(ROUNDS):
    for v in VARIANTS:
        table[v].append(timed(engines[v]))
for v in VARIANTS:
    t = table[v]
    print("variant %2d: %s   median %.2f us = %.3f Tsample/s" % (v, " ".join("%.2f" % x for x in t), float(np.median(t)),
                                                                N * R / float(np.median(t)) / 1e6))

# the board while the shipped kernel runs back to back: ~3 s of queued steps, two rocm-smi samples inside them
print("\nrocm-smi while the shipped C2 step runs back to back:")
steps(engines[0], 3000)
torch.cuda.synchronize()
steps(engines[0], 50000)
for _ in range(2):
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20)
        print("\n".join(l for l in r.stdout.split("\n") if any(k in l.lower() for k in ("power", "sclk", "mclk", "fclk", "temp"))))
    except Exception as exc:
        print("rocm-smi failed: %r" % (exc,))
    print("--")
torch.cuda.synchronize()
print("idle:")
time.sleep(1.0)
try:
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20)
    print("\n".join(l for l in r.stdout.split("\n") if any(k in l.lower() for k in ("power", "sclk", "mclk", "fclk"))))
except Exception as exc:
    print("rocm-smi failed: %r" % (exc,))
for ds in engines.values():
    ds.close()
