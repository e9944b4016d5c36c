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
returns None and callers say so ("fftw": "absent").  Round 6: where there is no FFTW but Intel MKL's FFTW3 interface
(libmkl_rt exports the same entry points), the same loop runs around THAT -- reported as "fftw3_api", never as FFTW.
"""
import ctypes
import ctypes.util
import time

import numpy as np

FFTW_FORWARD = -1          # fftw3.h
FFTW_MEASURE = 0
FFTW_ESTIMATE = 1 << 6

# (provider, soname).  Real FFTW first.  Then -- round 6, VERDICT r05 item 6 -- Intel MKL's FFTW3 INTERFACE: libmkl_rt
# exports fftwf_plan_dft_1d / fftwf_execute / ..., the exact API of /root/reference/src/datastore.cxx:30-33,82, and is what
# the reference links when a distribution's `fftw3f` is MKL's wrapper.  It is MKL's arithmetic (DFTI underneath), not
# FFTW's: it pins nothing and parity stays "unpinned"; it replaces "absent" with a float32 FFT nobody here wrote, run
# through the reference's own call sites, on every box that has it (this image: /opt/conda/lib).
_CANDIDATES = (("fftw3f", "libfftw3f.so.3"), ("fftw3f", "libfftw3f.so"),
               ("mkl_fftw3_interface", "libmkl_rt.so.2"), ("mkl_fftw3_interface", "libmkl_rt.so.1"),
               ("mkl_fftw3_interface", "libmkl_rt.so"),
               ("mkl_fftw3_interface", "/opt/conda/lib/libmkl_rt.so.2"), ("mkl_fftw3_interface", "/opt/conda/lib/libmkl_rt.so.1"),
               ("mkl_fftw3_interface", "/opt/conda/lib/libmkl_rt.so"))
_lib = False
_provider = None


def _bind(lib):
    lib.fftwf_malloc.restype = ctypes.c_void_p          # (fftwf_alloc_complex(n) = fftwf_malloc(8 n): datastore.cxx:30-31;
    lib.fftwf_malloc.argtypes = [ctypes.c_size_t]       #  FFTW < 3.3 and some wrappers lack the former)
    lib.fftwf_free.argtypes = [ctypes.c_void_p]
    lib.fftwf_plan_dft_1d.restype = ctypes.c_void_p
    lib.fftwf_plan_dft_1d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
    lib.fftwf_execute.argtypes = [ctypes.c_void_p]
    lib.fftwf_destroy_plan.argtypes = [ctypes.c_void_p]


def load(allow_mkl=True):
    """A library that implements the FFTW3 single-precision API, or None.  provider() says whose it is."""
    global _lib, _provider
    if _lib is False:
        _lib = None
        names = list(_CANDIDATES)
        found = ctypes.util.find_library("fftw3f")
        if found:
            names.insert(0, ("fftw3f", found))
        for provider, name in names:
            try:
                lib = ctypes.CDLL(name)
                _bind(lib)
            except (OSError, AttributeError):
                continue
            _lib, _provider = lib, provider
            break
    if _lib is not None and _provider != "fftw3f" and not allow_mkl:
        return None
    return _lib


def provider():
    """"fftw3f" (the reference's real provider), "mkl_fftw3_interface", or None."""
    load()
    return _provider


def accumulate(N, stream, repeats, window=None, flags=FFTW_MEASURE):
    """Datastore::fftThread (datastore.cxx:66-89) on one contiguous stream with real FFTW:
    returns (pwr[N] float64, repeats_done, seconds spent in the frame loop)."""
    lib = load()
    if lib is None:
        raise RuntimeError("no FFTW3-API library is available on this machine")
    inbuf = lib.fftwf_malloc(8 * N)                                      # datastore.cxx:30 (fftwf_alloc_complex)
    outbuf = lib.fftwf_malloc(8 * N)                                     # :31
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
    """For bench.py / tests: real-FFTW figures next to `others` (name -> spectrum of the same frames): max relative
    per-bin difference for both planner flags, and samples/s of the FFTW_MEASURE loop (numpy around fftwf_execute).
    "fftw" is "present" only for a real libfftw3f; with MKL's FFTW3 interface it stays "absent" and the same figures
    appear under "fftw3_api" with "provider": "mkl_fftw3_interface"; with neither, {"fftw": "absent"}."""
    if load() is None:
        return {"fftw": "absent", "fftw3_api": "absent"}
    real = provider() == "fftw3f"
    figures = {"provider": provider()}
    for label, flags in (("measure", FFTW_MEASURE), ("estimate", FFTW_ESTIMATE)):
        pwr, frames, seconds = accumulate(N, stream, repeats, window, flags)
        figures[label] = {"frames": frames, "samples_per_s": N * frames / max(seconds, 1e-12)}
        for name, other in others.items():
            if other is not None:
                figures[label]["max_rel_vs_" + name] = float(np.max(np.abs(np.asarray(other) - pwr) / pwr))
    if real:
        return dict(figures, fftw="present")
    return {"fftw": "absent", "fftw3_api": figures}
