import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtl_power_fftw_amd as rpf
dev = torch.device("cuda:0")
N, R = 4096, 10000
stream = rpf.synth.noise_tones_iq(2, N * R)
d_in = torch.from_numpy(stream).to(dev)
def ref(first, frames):
    x = d_in[2*N*first:2*N*(first+frames)].to(torch.float32).reshape(frames, N, 2) - 127.0
    sign = (1 - 2 * (torch.arange(N, device=dev) % 2)).to(torch.float32)
    z = torch.complex(x[..., 0] * sign, x[..., 1] * sign).to(torch.complex128)
    acc = torch.zeros(N, dtype=torch.float64, device=dev)
    for i in range(0, frames, 500):
        s = torch.fft.fft(z[i:i+500], dim=1)
        acc += (s.real ** 2 + s.imag ** 2).sum(0)
    return acc.cpu().numpy()
for vid, flags in ((0, 0), (1, 0), (0, 1), (2, 0)):
    ds = rpf.Datastore(rpf.Params(N=N, repeats=R), flags=(vid << 8) | flags)
    for first, frames in ((0, 10000), (0, 3333), (3333, 6667), (0, 768), (0, 769), (0, 1536), (0, 1537), (0, 2000)):
        out = torch.empty(N, dtype=torch.float64, device=dev)
        errs = []
        for rep in range(3):
            ds.accumulate_device(d_in.data_ptr() + 2 * N * first, 2 * N * frames, frames, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            g = out.cpu().numpy()
            errs.append(float(np.max(np.abs(g - ref(first, frames)) / ref(first, frames))))
        print("vid=%d flags=%d first=%d frames=%d  max rel err vs f64 over 3 runs: %s" % (vid, flags, first, frames, ["%.2e" % e for e in errs]), flush=True)
    ds.close()
