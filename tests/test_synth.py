"""The synthetic IQ source is bit-reproducible (integer-only).  (-m "not gpu")"""
import hashlib

import numpy as np

import rtl_power_fftw_amd as rpf


def test_splitmix64_reference_values():
    # first outputs of splitmix64 with seed 0 (the published test vector)
    got = rpf.synth.splitmix64(0, 3)
    assert [int(v) for v in got] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_streams_are_reproducible_and_chunk_independent():
    a = rpf.synth.noise_tones_iq(2, 100000)
    b = rpf.synth.noise_tones_iq(2, 100000, chunk=777)
    assert np.array_equal(a, b)
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
    assert a.dtype == np.uint8 and a.size == 200000
    assert 126.5 < a.mean() < 128.5 and 15 < a.std() < 30
    u = rpf.synth.uniform_iq(1, 1000)
    assert np.array_equal(u, rpf.synth.uniform_iq(1, 1000)) and not np.array_equal(u, rpf.synth.uniform_iq(2, 1000))


def test_hann_window():
    w = rpf.synth.hann_window(512)
    assert w.dtype == np.float32 and w[0] == 0 and abs(w[256] - 1) < 1e-7
    assert abs(np.mean(w.astype(np.float64) ** 2) - 0.375) < 1e-6


def test_torch_generator_is_bit_identical_to_numpy():
    """bench.py and the full-size GPU tests generate their streams with torch on the device;
    same bytes as the numpy generator the fixtures were made with, from any first sample."""
    import torch  # noqa: F401
    for seed, n, first in ((2, 100003, 0), (57, 65536, 12345), (4, 1 << 18, 262144 * 999)):
        a = rpf.synth.noise_tones_iq(seed, n, first=first)
        b = rpf.synth.noise_tones_iq_torch(seed, n, "cpu", first=first, chunk=70000).numpy()
        assert np.array_equal(a, b)
    whole = rpf.synth.noise_tones_iq(9, 50000)
    assert np.array_equal(whole[2 * 1234: 2 * 1234 + 2000], rpf.synth.noise_tones_iq(9, 1000, first=1234))
