"""Shared test helpers: ctypes view of the CPU oracle (the checker) and of the
host emulator, float64 truth, error metrics."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

u8p = ctypes.POINTER(ctypes.c_uint8)
dp = ctypes.POINTER(ctypes.c_double)
fp = ctypes.POINTER(ctypes.c_float)
i64p = ctypes.POINTER(ctypes.c_int64)


class PlanParams(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int), ("sample_rate", ctypes.c_int), ("repeats", ctypes.c_int64),
                ("integration_time_isSet", ctypes.c_int), ("integration_time", ctypes.c_double),
                ("buf_length", ctypes.c_int), ("buf_length_isSet", ctypes.c_int),
                ("freq_hopping_isSet", ctypes.c_int), ("startfreq", ctypes.c_int64),
                ("stopfreq", ctypes.c_int64), ("cfreq", ctypes.c_int64), ("min_overlap", ctypes.c_double)]


_oracle = None
_emul = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "librpf_oracle.so"))
        lib.rpf_oracle_accumulate.argtypes = [ctypes.c_int, fp, ctypes.c_int, u8p, ctypes.c_size_t,
                                              ctypes.c_int64, dp, i64p]
        lib.rpf_oracle_accumulate_mt.argtypes = [ctypes.c_int, fp, u8p, ctypes.c_size_t,
                                                 ctypes.c_int64, ctypes.c_int, dp, i64p]
        lib.rpf_oracle_worker_create.restype = ctypes.c_void_p
        lib.rpf_oracle_worker_create.argtypes = [ctypes.c_int, fp, ctypes.c_int]
        lib.rpf_oracle_worker_destroy.argtypes = [ctypes.c_void_p]
        lib.rpf_oracle_worker_begin.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        lib.rpf_oracle_worker_consume.argtypes = [ctypes.c_void_p, u8p, ctypes.c_size_t]
        lib.rpf_oracle_worker_repeats_done.restype = ctypes.c_int64
        lib.rpf_oracle_worker_repeats_done.argtypes = [ctypes.c_void_p]
        lib.rpf_oracle_worker_pwr.restype = dp
        lib.rpf_oracle_worker_pwr.argtypes = [ctypes.c_void_p]
        lib.rpf_oracle_plan_create.restype = ctypes.c_void_p
        lib.rpf_oracle_plan_create.argtypes = [ctypes.c_int]
        lib.rpf_oracle_plan_destroy.argtypes = [ctypes.c_void_p]
        lib.rpf_oracle_fft_f32.argtypes = [ctypes.c_void_p, fp, fp]
        lib.rpf_oracle_fft_f64.argtypes = [ctypes.c_void_p, dp, dp]
        lib.rpf_oracle_format_text.restype = ctypes.c_long
        lib.rpf_oracle_format_text.argtypes = [dp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_int, dp, ctypes.c_char_p,
                                               ctypes.c_size_t]
        lib.rpf_oracle_format_matrix.argtypes = [dp, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                                 ctypes.c_int, dp, fp]
        lib.rpf_oracle_make_plan.argtypes = [ctypes.POINTER(PlanParams), i64p, ctypes.c_int]
        lib.rpf_oracle_data_needed.restype = ctypes.c_int64
        lib.rpf_oracle_data_needed.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        _oracle = lib
    return _oracle


def emul_lib():
    global _emul
    if _emul is None:
        lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "librpf_emul.so"))
        lib.rpf_emul_accumulate.argtypes = [ctypes.c_int, ctypes.c_int, fp, u8p, ctypes.c_long, dp]
        lib.rpf_emul_bluestein.argtypes = [ctypes.c_int, fp, u8p, ctypes.c_long, dp]
        lib.rpf_emul_mixed.argtypes = [ctypes.c_int, fp, u8p, ctypes.c_long, dp]
        lib.rpf_emul_small_dft.argtypes = [ctypes.c_int, fp]
        lib.rpf_emul_shipped.argtypes = [ctypes.c_int, fp, u8p, ctypes.c_long, dp]
        _emul = lib
    return _emul


def emul_mixed(plan, stream, repeats, window=None):
    """Plan `plan` of tests/emul's list run through mixed_core.h thread by thread; returns (N, pwr)."""
    lib = emul_lib()
    N = lib.rpf_emul_mixed_n(plan)
    assert N > 0
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=np.float32)
        w = window.ctypes.data_as(fp)
    pwr = np.zeros(N)
    assert lib.rpf_emul_mixed(plan, w, stream.ctypes.data_as(u8p), repeats, pwr.ctypes.data_as(dp)) == 0
    return pwr


def emul_shipped_sizes():
    """The sizes of mixed_plans.inc + mixed_plans_split.inc, in table order."""
    lib = emul_lib()
    return [lib.rpf_emul_shipped_n(i) for i in range(lib.rpf_emul_shipped_count())]


def emul_shipped(index, stream, repeats, window=None):
    """Entry `index` of the shipped plan tables run through mixed_core.h thread by thread."""
    lib = emul_lib()
    N = lib.rpf_emul_shipped_n(index)
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=np.float32)
        w = window.ctypes.data_as(fp)
    pwr = np.full(N, np.nan)
    assert lib.rpf_emul_shipped(index, w, stream.ctypes.data_as(u8p), repeats, pwr.ctypes.data_as(dp)) == 0
    return pwr


def emul_mixed_n(plan):
    return emul_lib().rpf_emul_mixed_n(plan)


def emul_small_dft(x):
    """dft_small.h's in-register DFT of len(x) points."""
    v = np.empty(2 * len(x), np.float32)
    v[0::2], v[1::2] = x.real, x.imag
    assert emul_lib().rpf_emul_small_dft(len(x), v.ctypes.data_as(fp)) == 0
    return v[0::2] + 1j * v[1::2]


def emul_wide_dft(x):
    """dft_small_wide.h's double-precision DFT of len(x) points."""
    v = np.empty(2 * len(x), np.float64)
    v[0::2], v[1::2] = x.real, x.imag
    lib = emul_lib()
    lib.rpf_emul_wide_dft.argtypes = [ctypes.c_int, dp]
    assert lib.rpf_emul_wide_dft(len(x), v.ctypes.data_as(dp)) == 0
    return v[0::2] + 1j * v[1::2]


def emul_wide_pass(x):
    """dft_small_wide.h's fused wide pass of len(x) complex64 inputs: the outputs it hands on, as complex128."""
    v = np.empty(2 * len(x), np.float32)
    v[0::2], v[1::2] = x.real, x.imag
    out = np.zeros(2 * len(x), np.float64)
    lib = emul_lib()
    lib.rpf_emul_wide_pass.argtypes = [ctypes.c_int, fp, dp]
    assert lib.rpf_emul_wide_pass(len(x), v.ctypes.data_as(fp), out.ctypes.data_as(dp)) == 0
    return out[0::2] + 1j * out[1::2]


def oracle_accumulate(N, stream, repeats, window=None, precision=32):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    pwr = np.zeros(N)
    done = ctypes.c_int64()
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=np.float32)
        w = window.ctypes.data_as(fp)
    rc = oracle_lib().rpf_oracle_accumulate(N, w, precision, stream.ctypes.data_as(u8p), stream.size,
                                            repeats, pwr.ctypes.data_as(dp), ctypes.byref(done))
    assert rc == 0
    return pwr, done.value


class OracleWorker:
    """Buffer-by-buffer view of the oracle (Datastore::fftThread restated)."""

    def __init__(self, N, window=None, precision=32):
        self.N = N
        self._w = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
        self._h = oracle_lib().rpf_oracle_worker_create(
            N, None if self._w is None else self._w.ctypes.data_as(fp), precision)
        assert self._h

    def begin(self, repeats):
        oracle_lib().rpf_oracle_worker_begin(self._h, repeats)

    def consume(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        oracle_lib().rpf_oracle_worker_consume(self._h, buf.ctypes.data_as(u8p), buf.size)

    @property
    def repeats_done(self):
        return oracle_lib().rpf_oracle_worker_repeats_done(self._h)

    @property
    def pwr(self):
        return np.ctypeslib.as_array(oracle_lib().rpf_oracle_worker_pwr(self._h), shape=(self.N,)).copy()

    def close(self):
        if self._h:
            oracle_lib().rpf_oracle_worker_destroy(self._h)
            self._h = None


def emul_accumulate(N, P, stream, nframes, window=None):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    pwr = np.zeros(N)
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=np.float32)
        w = window.ctypes.data_as(fp)
    rc = emul_lib().rpf_emul_accumulate(N, P, w, stream.ctypes.data_as(u8p), nframes, pwr.ctypes.data_as(dp))
    assert rc == 0, "no emulator instantiation for N=%d P=%d" % (N, P)
    return pwr


def emul_bluestein(N, stream, nframes, window=None):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    pwr = np.zeros(N)
    w = None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=np.float32)
        w = window.ctypes.data_as(fp)
    rc = emul_lib().rpf_emul_bluestein(N, w, stream.ctypes.data_as(u8p), nframes, pwr.ctypes.data_as(dp))
    assert rc == 0, "no Bluestein emulation for N=%d" % N
    return pwr


def truth_f64(N, stream, repeats, window=None):
    """numpy complex128 evaluation of datastore.cxx:66-89 (exact unpack, float32
    window product, float64 FFT and accumulate)."""
    sign = (1 - 2 * (np.arange(N) % 2)).astype(np.float32)
    w = None if window is None else np.asarray(window, dtype=np.float32)
    total = np.zeros(N)
    chunk = max(1, (1 << 24) // N)                  # <= 16 M samples (0.8 GB of complex128 work arrays) at a time
    for f0 in range(0, repeats, chunk):
        f1 = min(repeats, f0 + chunk)
        x = np.asarray(stream[2 * N * f0: 2 * N * f1]).astype(np.float32).reshape(f1 - f0, N, 2) - np.float32(127.0)
        x = x * sign[None, :, None]
        if w is not None:
            x = x * w[None, :, None]
        z = x[..., 0].astype(np.float64) + 1j * x[..., 1].astype(np.float64)
        spec = np.fft.fft(z, axis=1)
        total += (spec.real ** 2 + spec.imag ** 2).sum(axis=0)
    return total


def max_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def max_err_over_mean(a, b):
    """Largest per-bin error in units of max(bin power, median bin power).  With
    only a few frames averaged some bins are orders of magnitude weaker than
    the typical bin (the periodogram of noise is exponentially distributed), and
    a float32 FFT's absolute error -- which scales with the frame's energy, not
    with the bin -- makes the per-bin *relative* error of those bins arbitrarily
    large in any float32 implementation, FFTW included (BASELINE.md
    2).  Bins at or above the median are still judged relative to themselves."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    floor = np.median(np.abs(b))
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: g[k] for k in g.files}


def golden_stream(g):
    import rtl_power_fftw_amd as rpf
    N, R = int(g["N"]), int(g["repeats"])
    gen = str(g["generator"])
    s = rpf.synth.uniform_iq(int(g["seed"]), N * R) if gen == "uniform" else rpf.synth.noise_tones_iq(int(g["seed"]), N * R)
    crc = np.uint64(int(np.bitwise_xor.reduce(s.view(np.uint8).astype(np.uint64) * np.arange(1, s.size + 1, dtype=np.uint64))))
    assert crc == g["stream_crc"], "synthetic generator is not reproducing the fixture's bytes"
    return s


# BASELINE.json configs[3] and [4] on their specified streams at full size (tests/golden/make_golden.py)
C4_CASE = "c4_n262144_r1000_seed4"
C5_CASES = ["c5_hop%d_n4096_r5000" % h for h in range(8)]


def harmonic_bins(N):
    """Bins where noise_tones_iq's two tones and the harmonics of their integer
    rounding land: all multiples of N/16 (the (-1)^n shift by N/2 maps the set onto itself)."""
    return np.arange(16) * (N // 16)


def golden_stream_device(g, dev):
    """The fixture's stream generated on the GPU (synth.noise_tones_iq_torch: the 0.5 GB of
    config C4 in milliseconds instead of a minute of numpy), verified against the fixture's
    checksum there.  Returns (host copy, device tensor)."""
    import torch

    import rtl_power_fftw_amd as rpf
    N, R = int(g["N"]), int(g["repeats"])
    assert str(g["generator"]) == "noise_tones"
    d = rpf.synth.noise_tones_iq_torch(int(g["seed"]), N * R, dev)
    acc = torch.zeros((), dtype=torch.int64, device=dev)
    step = 1 << 26
    for p0 in range(0, d.numel(), step):               # xor over bytes[i] * (i + 1), as golden_stream
        t = d[p0:p0 + step].to(torch.int64) * torch.arange(p0 + 1, p0 + 1 + min(step, d.numel() - p0),
                                                           dtype=torch.int64, device=dev)
        while t.numel() > 1:
            if t.numel() % 2:
                t = torch.cat([t, torch.zeros(1, dtype=torch.int64, device=dev)])
            half = t.numel() // 2
            t = t[:half] ^ t[half:]
        acc ^= t[0]
    assert np.uint64(int(acc.item()) & ((1 << 64) - 1)) == g["stream_crc"], \
        "the torch generator is not reproducing the fixture's bytes"
    return d.cpu().numpy(), d


GOLDEN_CASES = ["c1_n512_r100_uniform", "n512_r100_hann", "n4096_r64_noise", "n4096_r64_hann",
                "n64_r33_uniform", "n1024_r17_noise", "n8192_r9_noise", "n500_r20_uniform",
                "n262144_r2_uniform", "n16384_r8_uniform", "n5000_r12_uniform", "n100000_r3_uniform"]
