"""What the sizes that failed the held-out parity test do on the kernels that would take them back (large Bluestein,
the two-kernel four-step pair): the same streams, the same quantities.  Usage: python tools/gpu_heldout_alternatives.py N ...
(The held-out key stays in tests/test_gpu_heldout.py; this tool imports it and PICKS nothing.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtl_power_fftw_amd as rpf
from helpers import max_rel, oracle_accumulate, truth_f64
from test_gpu_heldout import tuning_stream_seeds
from test_gpu_parity import run_device

dev = torch.device("cuda:0")
F = rpf._lib
for N in [int(v) for v in sys.argv[1:]]:
    for name, seed in tuning_stream_seeds(N):
        stream = rpf.synth.noise_tones_iq(seed, N * 64)
        for windowed in (False, True):
            w = rpf.synth.hann_window(N) if windowed else None
            o32, _ = oracle_accumulate(N, stream, 64, w, 32)
            truth = truth_f64(N, stream, 64, w)
            row = []
            for label, flags in (("default", 0), ("no-mixed-radix", F.FLAG_NO_MIXED_RADIX),
                                 ("no-mixed, two-kernel", F.FLAG_NO_MIXED_RADIX | F.FLAG_NO_FOURSTEP_FUSED)):
                with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=64), w, flags=flags) as ds:
                    got, _ = run_device(ds, stream, 64, dev)
                row.append("%s: vs cpu %.2e vs truth %.2e" % (label, max_rel(got, o32), max_rel(got, truth)))
            print("N=%6d %s %s | cpu vs truth %.2e | %s" % (N, name, "hann" if windowed else "rect", max_rel(o32, truth), " | ".join(row)), flush=True)
