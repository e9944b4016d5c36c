"""Kernel-only timing (HIP events on the launch stream) + parity of every kernel
variant.  Usage: python tools/gpu_sweep.py [N:variant ...]
Variants other than 0 exist only in the tuning build: RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so
(make -C rtl-power-fftw_amd/csrc tuning).  SWEEP_NOWIN=1 skips the windowed cases, SWEEP_ONLYWIN=1 the plain ones;
SWEEP_K: timed launches per case (400); SWEEP_FLAGS: RPF_FLAG_* bits for every engine (2 fused four-step, 4 no mixed radix)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "librpf_oracle.so"))
c_u8p = ctypes.POINTER(ctypes.c_uint8); c_dp = ctypes.POINTER(ctypes.c_double); c_fp = ctypes.POINTER(ctypes.c_float)
orc.rpf_oracle_accumulate.argtypes = [ctypes.c_int, c_fp, ctypes.c_int, c_u8p, ctypes.c_size_t, ctypes.c_int64, c_dp, ctypes.POINTER(ctypes.c_int64)]

def oracle(N, buf, R, win=None, prec=64):
    pwr = np.zeros(N); done = ctypes.c_int64()
    w = win.ctypes.data_as(c_fp) if win is not None else None
    assert orc.rpf_oracle_accumulate(N, w, prec, buf.ctypes.data_as(c_u8p), buf.size, R, pwr.ctypes.data_as(c_dp), ctypes.byref(done)) == 0
    return pwr

dev = torch.device("cuda:0")
TOTAL = 4096 * 10000            # complex samples per launch (same bytes for every N)
cases = sys.argv[1:] or ["4096:0", "4096:1", "4096:2", "4096:3", "512:0", "512:1", "1024:0", "1024:1",
                         "2048:0", "2048:1", "8192:0", "64:0", "128:0", "256:0"]
NB = 8
bufs = [rpf.synth.noise_tones_iq_torch(2, TOTAL, dev)]
base = bufs[0].cpu().numpy()
bufs += [torch.roll(bufs[0], shifts=8192 * 37 * i) for i in range(1, NB)]
ref_full = {}
truth_cache = {}
s = torch.cuda.current_stream().cuda_stream
for case in cases:
    N, vid = (int(v) for v in case.split(":"))
    R = TOTAL // N
    for win in ((False,) if os.environ.get("SWEEP_NOWIN") else (True,) if os.environ.get("SWEEP_ONLYWIN") else (False, True)):
        w = rpf.synth.hann_window(N) if win else None
        try:
            ds = rpf.Datastore(rpf.Params(N=N, window=win, repeats=R), w, flags=(vid << 8) | int(os.environ.get("SWEEP_FLAGS", "0")))
        except rpf.RPFError as ex:
            print("N=%d v=%d win=%d: %s" % (N, vid, win, ex)); continue
        d_pwr = torch.zeros(N, dtype=torch.float64, device=dev)
        # parity on the first 64 frames
        RC = 64
        ds.accumulate_device(bufs[0].data_ptr(), 2 * N * RC, RC, d_pwr.data_ptr(), s)
        torch.cuda.synchronize()
        if (N, win) not in truth_cache:
            truth_cache[(N, win)] = oracle(N, base[: 2 * N * RC], RC, w)
        t = truth_cache[(N, win)]
        err = float(np.max(np.abs(d_pwr.cpu().numpy() - t) / t))
        # full-size result against variant 0's (the variants reorder work, not arithmetic)
        ds.accumulate_device(bufs[0].data_ptr(), 2 * N * R, R, d_pwr.data_ptr(), s)
        torch.cuda.synchronize()
        full = d_pwr.cpu().numpy().copy()
        if vid == 0:
            ref_full[(N, win)] = full
        same = "ref" if vid == 0 else ("n/a" if (N, win) not in ref_full else
                                       "max_rel_vs_v0 %.1e" % float(np.max(np.abs(full - ref_full[(N, win)]) / ref_full[(N, win)])))
        K = int(os.environ.get("SWEEP_K", "400"))
        for i in range(K // 2):
            ds.device_fused(bufs[i % NB].data_ptr(), 2 * N * R, R, s)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            ds.device_fused(bufs[i % NB].data_ptr(), 2 * N * R, R, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        e0.record()
        for i in range(K):
            ds.device_reduce(d_pwr.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        msr = e0.elapsed_time(e1) / K
        print("N=%5d v=%d win=%d  K1 %.4f ms  %.1f Gsample/s  %.0f GB/s (%.1f%% of 8 TB/s)  K3 %.4f ms  err-vs-f64 %.2e  %s  %s"
              % (N, vid, win, ms, TOTAL / ms / 1e6, 2 * TOTAL / ms / 1e6, 2 * TOTAL / ms / 1e6 / 80, msr, err, same, ds.launch_info()), flush=True)
        ds.close()
