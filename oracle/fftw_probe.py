"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product.

Run-time probe for the reference's real FFT provider.  rtl_power_fftw computes its
spectra with FFTW 3 single precision (`fftw3f`, version unpinned:
/root/reference/CMakeLists.txt:7; call sites /root/reference/src/datastore.cxx:30-33,82).
Neither the header nor the library exists in the build image, so the oracle
(rpf_oracle.c) stands its own FFT where FFTW stands and parity is "unpinned".  If a
box does have `libfftw3f.so.3`, this module dlopen()s it (ctypes; nothing links it) and
runs the reference's worker loop (datastore.cxx:66-89) around FFTW's own
`fftwf_execute`, with the reference's planner flag (FFTW_MEASURE, :32-33) and with
FFTW_ESTIMATE, so that tests and bench.py's cpu_baseline leg can report real-FFTW
spectra and timing next to the oracle's and the GPU's.  Absent library: `load()`
returns None and callers say so ("fftw": "absent").
"""
import ctypes
import ctypes.util
import time

import numpy as np

FFTW_FORWARD = -1          # fftw3.h
FFTW_MEASURE = 0
FFTW_ESTIMATE = 1 << 6

_CANDIDATES = ("libfftw3f.so.3", "libfftw3f.so")
_lib = False


def load():
    """The fftw3f library, or None when this box has none."""
    global _lib
    if _lib is not False:
        return _lib
    _lib = None
    names = list(_CANDIDATES)
    found = ctypes.util.find_library("fftw3f")
    if found:
        names.append(found)
    for name in names:
        try:
            lib = ctypes.CDLL(name)
        except OSError:
            continue
        try:
            lib.fftwf_alloc_complex.restype = ctypes.c_void_p
            lib.fftwf_alloc_complex.argtypes = [ctypes.c_size_t]
            lib.fftwf_free.argtypes = [ctypes.c_void_p]
            lib.fftwf_plan_dft_1d.restype = ctypes.c_void_p
            lib.fftwf_plan_dft_1d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_uint]
            lib.fftwf_execute.argtypes = [ctypes.c_void_p]
            lib.fftwf_destroy_plan.argtypes = [ctypes.c_void_p]
        except AttributeError:
            continue
        _lib = lib
        break
    return _lib


def accumulate(N, stream, repeats, window=None, flags=FFTW_MEASURE):
    """Datastore::fftThread (datastore.cxx:66-89) on one contiguous stream with real FFTW:
    returns (pwr[N] float64, repeats_done, seconds spent in the frame loop)."""
    lib = load()
    if lib is None:
        raise RuntimeError("libfftw3f is not available on this machine")
    inbuf = lib.fftwf_alloc_complex(N)                                   # datastore.cxx:30
    outbuf = lib.fftwf_alloc_complex(N)                                  # :31
    plan = lib.fftwf_plan_dft_1d(N, inbuf, outbuf, FFTW_FORWARD, flags)  # :32-33 (planning may clobber inbuf)
    x = np.ctypeslib.as_array(ctypes.cast(inbuf, ctypes.POINTER(ctypes.c_float)), shape=(N, 2))
    X = np.ctypeslib.as_array(ctypes.cast(outbuf, ctypes.POINTER(ctypes.c_float)), shape=(N, 2))
    sign = (1 - 2 * (np.arange(N) % 2)).astype(np.float32)[:, None]      # :69-72
    w = None if window is None else np.asarray(window, dtype=np.float32)[:, None]
    pwr = np.zeros(N)
    stream = np.asarray(stream, dtype=np.uint8)
    frames = min(int(repeats), stream.size // (2 * N))
    t0 = time.perf_counter()
    for f in range(frames):
        raw = stream[2 * N * f: 2 * N * (f + 1)].reshape(N, 2)
        v = (raw.astype(np.float32) - np.float32(127.0)) * sign          # :73-77
        x[:] = v if w is None else v * w
        lib.fftwf_execute(plan)                                          # :82
        Xd = X.astype(np.float64)
        pwr += Xd[:, 0] * Xd[:, 0] + Xd[:, 1] * Xd[:, 1]                 # :83-85
    seconds = time.perf_counter() - t0
    lib.fftwf_destroy_plan(plan)
    lib.fftwf_free(inbuf)
    lib.fftwf_free(outbuf)
    return pwr, frames, seconds


def report(N, stream, repeats, others, window=None):
    """For bench.py / tests: {"fftw": "absent"} or real-FFTW figures next to `others`
    (name -> spectrum of the same frames): max relative per-bin difference for both
    planner flags, and samples/s of the FFTW_MEASURE loop (numpy around fftwf_execute)."""
    if load() is None:
        return {"fftw": "absent"}
    out = {"fftw": "present"}
    for label, flags in (("measure", FFTW_MEASURE), ("estimate", FFTW_ESTIMATE)):
        pwr, frames, seconds = accumulate(N, stream, repeats, window, flags)
        out[label] = {"frames": frames, "samples_per_s": N * frames / max(seconds, 1e-12)}
        for name, other in others.items():
            out[label]["max_rel_vs_" + name] = float(np.max(np.abs(np.asarray(other) - pwr) / pwr))
    return out
