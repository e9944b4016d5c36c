"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product.

A third float32 FFT of independent provenance beside the oracle's own (rpf_oracle_fft.inc) and, where a box
has it, real FFTW (fftw_probe.py): Intel MKL's DFTI, reached through `torch.fft.fft` on CPU complex64 tensors
(this image's torch is built against MKL 2024.2 -- `torch.__config__.show()`; MKL is also what the reference
links when its `fftw3f` is MKL's FFTW3 interface wrapper).  The reference's worker loop (datastore.cxx:66-89)
is run around it: exact unpack, (-1)^n, single-rounded window product, float32 transform, squares and sums in
double.  It does not pin the oracle to FFTW -- nothing short of FFTW can -- but it puts the oracle, the GPU and
an industrial float32 FFT side by side on every box, including the ones without FFTW.
"""
import numpy as np


def available():
    try:
        import torch
        return "mkl" in torch.__config__.show().lower() and torch.backends.mkl.is_available()
    except Exception:
        return False


def accumulate(N, stream, repeats, window=None, chunk=256):
    import torch
    stream = np.asarray(stream, dtype=np.uint8)
    frames = min(int(repeats), stream.size // (2 * N))
    sign = torch.from_numpy((1 - 2 * (np.arange(N) % 2)).astype(np.float32))
    w = None if window is None else torch.from_numpy(np.asarray(window, dtype=np.float32))
    pwr = torch.zeros(N, dtype=torch.float64)
    for f0 in range(0, frames, chunk):
        f1 = min(frames, f0 + chunk)
        raw = torch.from_numpy(stream[2 * N * f0: 2 * N * f1].reshape(f1 - f0, N, 2).astype(np.float32))
        v = (raw - 127.0) * sign[None, :, None]                         # datastore.cxx:73-77 (exact in float32)
        if w is not None:
            v = v * w[None, :, None]                                    # one rounding, as the reference
        X = torch.fft.fft(torch.view_as_complex(v.contiguous()), dim=1)  # complex64 in, complex64 out (MKL DFTI)
        assert X.dtype == torch.complex64
        Xd = torch.view_as_real(X).to(torch.float64)
        pwr += (Xd[..., 0] * Xd[..., 0] + Xd[..., 1] * Xd[..., 1]).sum(dim=0)   # :83-85
    return pwr.numpy(), frames


def report(N, stream, repeats, others, window=None):
    if not available():
        return {"mkl": "absent"}
    pwr, frames = accumulate(N, stream, repeats, window)
    out = {"mkl": "present", "frames": frames}
    for name, other in others.items():
        if other is not None:
            out["max_rel_vs_" + name] = float(np.max(np.abs(np.asarray(other) - pwr) / pwr))
    return out
