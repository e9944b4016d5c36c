"""Score kernel variants by what the parity tests measure: GPU against the CPU path (float32 oracle), plain per-bin
max-rel, 64 frames of the tone stream -- on TWO streams (the tone-stream parity test's own and another seed), windowed and
not, plus a short kernel-only timing.  Usage: python tools/gpu_parity_score.py N:variant ...
(variants other than 0: RPF_ENGINE_LIB=rtl-power-fftw_amd/librpf_engine_tuning.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "librpf_oracle.so"))
c_u8p = ctypes.POINTER(ctypes.c_uint8); c_dp = ctypes.POINTER(ctypes.c_double); c_fp = ctypes.POINTER(ctypes.c_float)
orc.rpf_oracle_accumulate.argtypes = [ctypes.c_int, c_fp, ctypes.c_int, c_u8p, ctypes.c_size_t, ctypes.c_int64, c_dp, ctypes.POINTER(ctypes.c_int64)]


def oracle32(N, buf, R, win):
    pwr = np.zeros(N); done = ctypes.c_int64()
    w = win.ctypes.data_as(c_fp) if win is not None else None
    assert orc.rpf_oracle_accumulate(N, w, 32, buf.ctypes.data_as(c_u8p), buf.size, R, pwr.ctypes.data_as(c_dp), ctypes.byref(done)) == 0
    return pwr


dev = torch.device("cuda:0")
R = 64
cache = {}
s = torch.cuda.current_stream().cuda_stream
for case in sys.argv[1:]:
    N, vid = (int(v) for v in case.split(":"))
    seeds = (300 + N % 89, 1300 + N % 97)          # [0]: test_tone_stream_parity_where_the_margin_is_thin's
    for win in (False, True):
        w = rpf.synth.hann_window(N) if win else None
        try:
            ds = rpf.Datastore(rpf.Params(N=N, window=win, repeats=R), w, flags=(vid << 8))
        except rpf.RPFError as ex:
            print("N=%d v=%d win=%d: %s" % (N, vid, win, ex)); continue
        errs = []
        for seed in seeds:
            key = (N, win, seed)
            if key not in cache:
                stream = rpf.synth.noise_tones_iq(seed, N * R)
                cache[key] = (torch.from_numpy(stream).to(dev), oracle32(N, stream, R, w))
            d_in, ref = cache[key]
            d_pwr = torch.zeros(N, dtype=torch.float64, device=dev)
            ds.accumulate_device(d_in.data_ptr(), 2 * N * R, R, d_pwr.data_ptr(), s)
            torch.cuda.synchronize()
            errs.append(float(np.max(np.abs(d_pwr.cpu().numpy() - ref) / ref)))
        # timing: 40 launches of ~82 MB (frames repeated from the cached 64)
        reps = max(1, (4096 * 10000) // (N * R))
        big = cache[(N, win, seeds[0])][0].repeat(reps)
        FR = R * reps
        for i in range(10):
            ds.device_fused(big.data_ptr(), 2 * N * FR, FR, s)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ds.device_fused(big.data_ptr(), 2 * N * FR, FR, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print("N=%d v=%d win=%d  %.1f Gsample/s  vs-oracle %.2e %.2e" % (N, vid, win, N * FR / ms / 1e6, errs[0], errs[1]), flush=True)
        ds.close()
        del big
