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
d_in = (torch.from_numpy(rpf.synth.uniform_iq(4, N * R)).to(dev) if os.environ.get("PROFILE_STREAM") == "uniform"
        else rpf.synth.noise_tones_iq_torch(4, N * R, dev))      # PROFILE_STREAM=uniform: uniformly random bytes instead of noise + tones
SHIFT = int(os.environ.get("PROFILE_SHIFT", "0"))        # the stream this many bytes further into its allocation (multiple of 16)
if SHIFT:
    d_big = torch.empty(d_in.numel() + SHIFT, dtype=torch.uint8, device=dev)
    d_big[SHIFT:] = d_in
    d_in = d_big[SHIFT:]
d_out = torch.zeros(N, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
names = ["P samples -> registers, barrier, next rows' DMA issue", "P both column groups", "P wait: buffer free", "P stores + drain", "P role barrier + arrive", "-", "-",
         "C wait: round produced", "C tile load -> LDS", "C role barrier + arrive", "C row FFTs + acc", "-", "-", "-"]
for flags, label in ((rpf._lib.FLAG_FOURSTEP_FUSED, "fused"), (rpf._lib.FLAG_NO_MIXED_RADIX | rpf._lib.FLAG_NO_FOURSTEP_FUSED, "two-kernel")):
    with rpf.Datastore(rpf.Params(N=N, repeats=R), flags=flags) as ds:
        lib = ds._lib
        if hasattr(lib, "rpf_debug_fused_knobs") and os.environ.get("RPF_FUSED_KNOBS"):
            kn = (ctypes.c_int * 4)(*[int(v) for v in os.environ["RPF_FUSED_KNOBS"].split(",")])
            lib.rpf_debug_fused_knobs(kn)
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
        if flags == rpf._lib.FLAG_FOURSTEP_FUSED and hasattr(lib, "rpf_debug_fused_profile"):
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

        if flags == rpf._lib.FLAG_FOURSTEP_FUSED and hasattr(lib, "rpf_debug_fused_trace"):
            tr = np.zeros((256, 64, 4), dtype=np.uint64)
            lib.rpf_debug_fused_trace(tr.ctypes.data_as(ctypes.c_void_p))
            tr = tr.astype(np.int64).reshape(8, 32, 64, 4) / 100.0        # microseconds; [xcd][rank][round][event]
            x = tr[0]                                                       # team 0 (the trace is of the LAST launch)
            rounds_ = range(10, 40)
            pa = x[:, 10:40, 0]; cs = x[:, 10:40, 1]; ca = x[:, 10:40, 2]; ps = x[:, 10:40, 3]
            print("  hand-offs of team 0, rounds 10..39, microseconds (mean over rounds):")
            print("   producers' arrivals, last - first CU           %6.2f" % np.mean(pa.max(0) - pa.min(0)))
            print("   last producer arrival -> first consumer sees   %6.2f   -> last consumer sees %6.2f" % (np.mean(cs.min(0) - pa.max(0)), np.mean(cs.max(0) - pa.max(0))))
            print("   consumers' arrivals, last - first CU           %6.2f" % np.mean(ca.max(0) - ca.min(0)))
            print("   consumer sees -> consumer arrives (mean CU)    %6.2f" % np.mean(ca - cs))
            print("   last consumer arrival -> first producer sees   %6.2f   -> last producer sees %6.2f" % (np.mean(ps[:, 1:].min(0) - ca[:, :-1].max(0)), np.mean(ps[:, 1:].max(0) - ca[:, :-1].max(0))))
            print("   producer sees -> producer arrives (mean CU)    %6.2f" % np.mean(pa[:, 1:] - ps[:, 1:]))
            print("   round period                                   %6.2f" % np.mean(np.diff(pa.max(0))))
            # which round do the consumers work on while the producers arrive for round j?  (lag in rounds, from the stamps)
            lag_c = np.mean(cs.mean(0) - pa.max(0))
            print("   consumers see round j this long after its last producer arrival (mean CU) %6.2f" % lag_c)
            print("   producers see `consumed` of round j - NBUF this long BEFORE/after they finish transforms: see 'P wait: buffer free'")
            # per workgroup (rank = 32 / TPF frame slot x tile): how long each takes from seeing `produced` to arriving on
            # `consumed` (tile load + barrier), from seeing `consumed` to arriving on `produced` (stores + drain + barrier),
            # and how far behind the team's first it runs
            print("   per rank: consumer sees->arrives | producer sees->arrives | consumer's lag behind the team's first to see")
            for r in range(32):
                print("    rank %2d: %6.2f | %6.2f | %6.2f" % (r, np.mean(ca[r] - cs[r]), np.mean(pa[r, 1:] - ps[r, 1:]), np.mean(cs[r] - cs.min(0))))
            if hasattr(lib, "rpf_debug_fused_profile_wg"):
                pw = np.zeros((256, 16), dtype=np.uint64)
                lib.rpf_debug_fused_profile_wg(pw.ctypes.data_as(ctypes.c_void_p))
                pw = pw.astype(np.float64).reshape(8, 32, 16) / rounds / 100.0      # us per round, [xcd][rank][segment]
                print("   per rank of team 0, us per round: P top | P transforms | P wait | P stores+drain | P barrier || C wait | C tile load | C barrier | C transforms")
                for r in range(32):
                    print("    rank %2d: " % r + " ".join("%5.2f" % pw[0, r, i] for i in range(5)) + " || " + " ".join("%5.2f" % pw[0, r, 8 + i] for i in range(4)))
            for team in range(8):
                y = tr[team]
                lagc = np.mean(y[:, 10:40, 1] - y[:, 10:40, 1].min(0), axis=1)
                lagp = np.mean(y[:, 10:40, 0] - y[:, 10:40, 0].min(0), axis=1)
                print("   team %d: consumers' lag by rank " % team + " ".join("%.0f" % v for v in lagc) + " | producers' arrival lag " + " ".join("%.0f" % v for v in lagp))
