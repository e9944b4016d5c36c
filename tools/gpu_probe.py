"""First-contact GPU probe: parity of every kernel variant vs the oracle, the
streaming path with straddling buffers, and a rough timing of config C2."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "librpf_oracle.so"))
c_u8p = ctypes.POINTER(ctypes.c_uint8); c_dp = ctypes.POINTER(ctypes.c_double); c_fp = ctypes.POINTER(ctypes.c_float)
orc.rpf_oracle_accumulate.argtypes = [ctypes.c_int, c_fp, ctypes.c_int, c_u8p, ctypes.c_size_t, ctypes.c_int64, c_dp, ctypes.POINTER(ctypes.c_int64)]

def oracle(N, buf, R, win=None, prec=32):
    pwr = np.zeros(N); done = ctypes.c_int64()
    w = win.ctypes.data_as(c_fp) if win is not None else None
    rc = orc.rpf_oracle_accumulate(N, w, prec, buf.ctypes.data_as(c_u8p), buf.size, R, pwr.ctypes.data_as(c_dp), ctypes.byref(done))
    assert rc == 0
    return pwr, done.value

def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))

print("device:", torch.cuda.get_device_name(0), "hip", torch.version.hip, flush=True)
dev = torch.device("cuda:0")
ok = True
for N in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    R = 37 + (4096 // N) * 13
    buf = rpf.synth.uniform_iq(N, N * R)
    win = rpf.synth.hann_window(N)
    d_buf = torch.from_numpy(buf).to(dev)
    for w in (None, win):
        p32, _ = oracle(N, buf, R, w, 32)
        p64, _ = oracle(N, buf, R, w, 64)
        for flags in (0, rpf._lib.FLAG_NO_LDS_DMA):
            params = rpf.Params(N=N, window=w is not None, repeats=R)
            try:
                ds = rpf.Datastore(params, w, flags=flags)
                d_pwr = torch.full((N,), -1.0, dtype=torch.float64, device=dev)
                n = ds.accumulate_device(d_buf.data_ptr(), buf.size, R, d_pwr.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                g = d_pwr.cpu().numpy()
                e32, e64 = relerr(g, p32), relerr(g, p64)
                # streaming path, buffer size that makes frames straddle
                ps, done = ds.accumulate(buf, R)
                es = relerr(ps, g)
                good = n == R and done == R and e64 < 5e-7 and e32 < 1e-6 and es < 1e-12
                ok &= good
                print("N=%5d win=%d dma=%d frames=%d  gpu-vs-oracle32 %.2e  gpu-vs-f64 %.2e  (oracle32-vs-f64 %.2e)  stream-vs-dev %.1e %s %s"
                      % (N, w is not None, flags == 0, n, e32, e64, relerr(p32, p64), es, ds.launch_info(), "OK" if good else "FAIL"), flush=True)
                ds.close()
            except Exception as ex:
                ok = False
                print("N=%d win=%s flags=%d EXC %r" % (N, w is not None, flags, ex), flush=True)

# straddling: small odd-sized buffers
N, R = 4096, 300
buf = rpf.synth.noise_tones_iq(7, N * R + 1000)
p32, d32 = oracle(N, buf, R - 3, None, 32)
for bl in (16384, 16384 * 3, 10000, 8190 * 2 + 2):
    ds = rpf.Datastore(rpf.Params(N=N, buf_length=bl, repeats=R - 3, buffers=3))
    ps, done = ds.accumulate(buf, R - 3)
    print("straddle buf_length=%d done=%d err=%.2e hist=%s" % (bl, done, relerr(ps, p32), ds.queue_histogram), flush=True)
    ok &= (done == R - 3) and relerr(ps, p32) < 1e-6
    ds.close()

# timing, config C2: N=4096, 10000 frames
N, R = 4096, 10000
NB = 6
bufs = [torch.from_numpy(rpf.synth.noise_tones_iq(100 + i, N * R)).to(dev) for i in range(NB)]
for flags in (0, rpf._lib.FLAG_NO_LDS_DMA):
    ds = rpf.Datastore(rpf.Params(N=N, repeats=R), flags=flags)
    d_pwr = torch.zeros(N, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for i in range(5):
        ds.accumulate_device(bufs[i % NB].data_ptr(), 2 * N * R, R, d_pwr.data_ptr(), s)
    torch.cuda.synchronize()
    K = 50
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
        ds.accumulate_device(bufs[i % NB].data_ptr(), 2 * N * R, R, d_pwr.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("C2 dma=%d: %.3f ms/step  %.1f Gsample/s  %.1f GB/s  %s" % (flags == 0, ms, N * R / ms / 1e6, 2 * N * R / ms / 1e6, ds.launch_info()), flush=True)
    ds.close()
print("PROBE", "PASS" if ok else "FAIL")
