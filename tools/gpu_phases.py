"""Per-phase cycle breakdown of K1 (needs the -DRPF_PHASE_TIMING build:
RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_timing.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10000 * 4096 // N
dev = torch.device("cuda:0")
lib = rpf.load()
dbg = lib.rpf_debug_phase_cycles
dbg.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = torch.from_numpy(rpf.synth.noise_tones_iq(2, N * R)).to(dev)
ds = rpf.Datastore(rpf.Params(N=N, repeats=R), flags=(vid << 8))
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ds.device_fused(buf.data_ptr(), 2 * N * R, R, s)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)(); waves = ctypes.c_ulonglong()
dbg(out, ctypes.byref(waves), 1)
K = 10
for _ in range(K):
    ds.device_fused(buf.data_ptr(), 2 * N * R, R, s)
torch.cuda.synchronize()
dbg(out, ctypes.byref(waves), 1)
info = ds.launch_info()
frames_per_wave = R * K / (waves.value / (info["block"] // 64) * info["frames_per_wg"] / K) / K if waves.value else 0
names = {0: "wait staged bytes", 1: "unpack", 2: "top barrier", 3: "DMA issue", 4: "p1: (fetch)", 5: "p1: butterfly+tw", 6: "p1: store",
         7: "p1: sync", 8: "p2: fetch", 9: "p2: butterfly+tw", 10: "p2: store", 11: "p2: sync", 12: "last fetch",
         13: "last butterfly", 14: "accumulate"}
tot = sum(out)
nw = waves.value
rounds = R / info["grid"] / info["frames_per_wg"]
print("N=%d v=%d R=%d waves=%d rounds/wave=%.2f  total cycles/wave/round = %.0f" % (N, vid, R, nw, rounds, tot / nw / rounds))
for i in range(16):
    if out[i]:
        print("  %-26s %8.0f cycles/round  %5.1f%%" % (names.get(i, "slot %d" % i), out[i] / nw / rounds, 100.0 * out[i] / tot))
ds.close()
