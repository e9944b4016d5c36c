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
names = ["P samples -> registers, barrier, next rows' DMA issue", "P both column groups", "P wait: buffer free", "P stores + drain", "P role barrier + arrive", "-", "-",
         "C wait: round produced", "C tile load -> LDS", "C role barrier + arrive", "C row FFTs + acc", "-", "-", "-"]
for flags, label in ((rpf._lib.FLAG_FOURSTEP_FUSED, "fused"), (0, "two-kernel")):
    with rpf.Datastore(rpf.Params(N=N, repeats=R), flags=flags) as ds:
        lib = ds._lib
        for _ in range(3):
            ds.accumulate_device(d_in.data_ptr(), 2 * N * R, R, d_out.data_ptr(), s)
        torch.cuda.synchronize()
        prof = (ctypes.c_ulonglong * 16)()
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
            rounds = (R * N // 262144 + 7) // 8
            for base, role in ((0, "producers"), (8, "consumers")):
                wgs = max(1, prof[base + 7])
                tot = sum(prof[base + i] for i in range(7))
                print("  %s, per workgroup and round (%d rounds per team), microseconds:" % (role, rounds))
                for i in range(7):
                    if names[(base and 7) + i] != "-":
                        print("   %-56s %8.3f  (%4.1f %%)" % (names[(base and 7) + i], prof[base + i] / wgs / rounds / 100.0, 100.0 * prof[base + i] / max(1, tot)))
                print("   total %.3f us per round" % (tot / wgs / rounds / 100.0))
