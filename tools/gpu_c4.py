"""Config C4 at full size on the GPU box: N = 262144 bins x 1000 repeats (524 MB of
IQ resident in HBM), four-step kernels K2a/K2b + K3.  Prints one JSON line (whole-
acquisition throughput, algorithmic GB/s) -- run under rocprofv3 by
tools/gpu_profile.sh for the per-kernel table in profiles/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtl_power_fftw_amd as rpf

N, R = 262144, 1000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
stream = rpf.synth.noise_tones_iq(4, N * R)
d_in = torch.from_numpy(stream).to(dev)
d_out = torch.zeros(N, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
    for _ in range(3):
        ds.accumulate_device(d_in.data_ptr(), stream.size, R, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ds.accumulate_device(d_in.data_ptr(), stream.size, R, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pwr = d_out.cpu().numpy()
x = stream.astype(np.int64).reshape(-1, 2) - 127
parseval = float(pwr.sum() / (float(N) * float(np.sum(x * x))))
print(json.dumps({"workload": "C4: N=262144 bins x 1000 repeats, rectangular window, u8 IQ resident in HBM",
                  "ms_per_acquisition": dt * 1e3, "samples_per_s": N * R / dt,
                  "algorithmic_GBps": (2.0 * N * R + 8 * N) / dt / 1e9,
                  "frac_of_8TBps": (2.0 * N * R + 8 * N) / dt / 8e12, "parseval_ratio": parseval}))
