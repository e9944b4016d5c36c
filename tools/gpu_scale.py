"""Kernel-only time of the N=4096 fused kernel vs frames per launch: separates the
per-workgroup fixed cost (twiddle load, first DMA, flush) from the per-frame cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_power_fftw_amd as rpf
dev = torch.device("cuda:0")
N = 4096
vid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
RMAX = 160000
g = torch.Generator(device=dev); g.manual_seed(1)
buf = torch.randint(96, 160, (2 * N * RMAX,), dtype=torch.uint8, device=dev, generator=g)
s = torch.cuda.current_stream().cuda_stream
ds = rpf.Datastore(rpf.Params(N=N, repeats=RMAX), flags=(vid << 8))
for R in (768, 1536, 2304, 5000, 10000, 20000, 40000, 80000, 160000):
    for i in range(3):
        ds.device_fused(buf.data_ptr(), 2 * N * R, R, s)
    torch.cuda.synchronize()
    K = 20
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(K):
        ds.device_fused(buf.data_ptr(), 2 * N * R, R, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("v=%d R=%6d  %.4f ms  %.1f Gsample/s  %.3f us/frame/WG-slot  %s" % (vid, R, ms, N * R / ms / 1e6, ms * 1e3 / (R / ds.launch_info()['grid']), ds.launch_info()), flush=True)
ds.close()
