"""Where a round of the fused four-step kernel spends its cycles (build with -DRPF_FUSED_PROFILE,
RPF_ENGINE_LIB=<that build>).  Usage: python tools/gpu_fused_profile.py [N] [frames]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
d_in = rpf.synth.noise_tones_iq_torch(4, N * R, dev)
d_out = torch.zeros(N, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
names = ["stage raw", "column FFTs", "wait prev Y read", "Y write+drain", "barrier 1", "Y tile load", "row FFTs+acc"]
for flags, label in ((rpf._lib.FLAG_FOURSTEP_FUSED, "fused"), (0, "two-kernel")):
    with rpf.Datastore(rpf.Params(N=N, repeats=R), flags=flags) as ds:
        lib = ds._lib
        for _ in range(3):
            ds.accumulate_device(d_in.data_ptr(), 2 * N * R, R, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = (ctypes.c_ulonglong * 10)()
        if hasattr(lib, "rpf_debug_fused_profile"):
            lib.rpf_debug_fused_profile(prof, 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        K = 10
        e0.record()
        for _ in range(K):
            ds.accumulate_device(d_in.data_ptr(), 2 * N * R, R, d_out.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        print("%s: %.3f ms per %d frames = %.1f Gsample/s" % (label, ms, R, N * R / ms / 1e6))
        if flags and hasattr(lib, "rpf_debug_fused_profile"):
            lib.rpf_debug_fused_profile(prof, 1)
            wgs = max(1, prof[9])
            rounds = (R * N // 262144 + 7) // 8
            tot = sum(prof[i] for i in range(7))
            print("  per workgroup and round (%d rounds per team), cycles:" % rounds)
            for i, n in enumerate(names):
                print("   %-18s %8.0f  (%4.1f %%)" % (n, prof[i] / wgs / rounds, 100.0 * prof[i] / max(1, tot)))
            print("   total %.0f cycles per round" % (tot / wgs / rounds))
