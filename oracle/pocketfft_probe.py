"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product.

A fourth float32 FFT of independent provenance beside the oracle's own (rpf_oracle_fft.inc), real FFTW where a box has
it (fftw_probe.py) and Intel MKL (mkl_probe.py): pocketfft, through `scipy.fft.fft`, which transforms complex64 input
IN single precision (numpy's own fft would promote to double).  The reference's worker loop (datastore.cxx:66-89) is
run around it: exact unpack, (-1)^n, single-rounded window product, float32 transform, squares and sums in double.
Like the MKL probe it pins nothing to FFTW; it puts one more float32 FFT nobody here wrote beside the oracle and the
GPU on every box.
"""
import numpy as np


def available():
    try:
        import scipy.fft  # noqa: F401
        return True
    except Exception:
        return False


def accumulate(N, stream, repeats, window=None):
    import scipy.fft
    stream = np.asarray(stream, dtype=np.uint8)
    frames = min(int(repeats), stream.size // (2 * N))
    sign = (1 - 2 * (np.arange(N) % 2)).astype(np.float32)
    if window is not None:
        sign = sign * np.asarray(window, dtype=np.float32)              # (w * +-1: exact; one rounding below, as the reference)
    pwr = np.zeros(N)
    for f in range(frames):
        raw = stream[2 * N * f: 2 * N * (f + 1)].reshape(N, 2).astype(np.float32) - np.float32(127.0)   # datastore.cxx:73-77
        z = (raw[:, 0] * sign + 1j * (raw[:, 1] * sign)).astype(np.complex64)
        X = scipy.fft.fft(z)
        assert X.dtype == np.complex64
        pwr += X.real.astype(np.float64) ** 2 + X.imag.astype(np.float64) ** 2                         # :83-85
    return pwr, frames


def report(N, stream, repeats, others, window=None):
    if not available():
        return {"pocketfft": "absent"}
    pwr, frames = accumulate(N, stream, repeats, window)
    out = {"pocketfft": "present (scipy.fft, single precision)", "frames": frames}
    for name, other in others.items():
        if other is not None:
            out["max_rel_vs_" + name] = float(np.max(np.abs(np.asarray(other) - pwr) / pwr))
    return out
