"""The gfx950 kernel's per-thread code (fft_core.h) executed thread by thread on
the host: checks index maps, twiddles, butterflies and unpack arithmetic of every
(N, P) instantiation against the oracle without a GPU.  (-m "not gpu")"""
import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import (emul_accumulate, emul_bluestein, emul_mixed, emul_mixed_n, emul_shipped, emul_shipped_sizes, emul_small_dft,
                     max_err_over_mean, max_rel,
                     oracle_accumulate, truth_f64)

CASES = [(64, 8), (128, 8), (256, 8), (512, 8), (1024, 8), (4096, 8), (128, 16), (256, 16), (512, 16),
         (1024, 16), (2048, 16), (4096, 16), (8192, 16)]


@pytest.mark.parametrize("N,P", CASES)
@pytest.mark.parametrize("windowed", [False, True])
def test_emulated_kernel_matches_oracle(N, P, windowed):
    R = 16
    stream = rpf.synth.noise_tones_iq(N + P, N * R)
    w = rpf.synth.hann_window(N) + np.float32(0.125) if windowed else None
    got = emul_accumulate(N, P, stream, R, w)
    t = truth_f64(N, stream, R, w)
    o32, _ = oracle_accumulate(N, stream, R, w, 32)
    assert max_rel(got, t) < 1e-6          # few frames -> little averaging of the float32 FFT noise
    assert max_rel(got, o32) < 1e-6        # north_star's parity bar


@pytest.mark.parametrize("N,P", [(512, 8), (4096, 16)])
def test_emulated_kernel_bin_placement(N, P):
    # asymmetric known answer: a single on-bin tone must land in exactly one bin
    k0 = 37
    n = np.arange(N)
    tone = 50.0 * np.exp(2j * np.pi * k0 * n / N)
    frame = np.empty(2 * N, dtype=np.uint8)
    frame[0::2] = np.rint(127 + tone.real)
    frame[1::2] = np.rint(127 + tone.imag)
    got = emul_accumulate(N, P, frame, 1)
    assert int(np.argmax(got)) == (k0 + N // 2) % N
    assert np.sort(got)[-2] < 1e-3 * got.max()


@pytest.mark.parametrize("N", [2, 6, 30, 32, 100, 500, 1000, 1536, 2046, 3000, 4094])
@pytest.mark.parametrize("windowed", [False, True])
def test_emulated_bluestein_kernel_matches_oracle(N, windowed):
    """Sizes that are not powers of two (the man page's -b 500 among them) go
    through Bluestein's chirp convolution on a power-of-two transform."""
    R = 32
    stream = rpf.synth.uniform_iq(N, N * R)
    w = rpf.synth.hann_window(N) + np.float32(0.25) if windowed else None
    got = emul_bluestein(N, stream, R, w)
    assert max_rel(got, truth_f64(N, stream, R, w)) < 1e-6
    o32, _ = oracle_accumulate(N, stream, R, w, 32)
    assert max_rel(got, o32) < 1e-6


@pytest.mark.parametrize("R", list(range(2, 26)))
def test_small_dft_radices(R):
    """dft_small.h: the in-register DFTs the mixed-radix kernel uses as radices (prime-factor maps,
    Cooley-Tukey steps with constexpr twiddles) against numpy's double DFT, and one basis vector per
    output to pin the index maps."""
    rng = np.random.default_rng(R)
    x = (rng.normal(size=R) + 1j * rng.normal(size=R)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(emul_small_dft(x) - ref).max() < 3e-7 * np.abs(ref).max()
    for k in range(R):
        tone = np.exp(2j * np.pi * k * np.arange(R) / R).astype(np.complex64)
        got = emul_small_dft(tone)
        assert int(np.argmax(np.abs(got))) == k and abs(got[k] - R) < 1e-5 * R
        assert np.abs(np.delete(got, k)).max() < 2e-6 * R


@pytest.mark.parametrize("R", list(range(2, 26)))
def test_wide_dft_radices(R):
    """dft_small_wide.h: every radix once more in double (the split forms' wide last passes, round 5) against numpy's
    double DFT to double precision, natural order in and out like SmallDft -- the accumulators the outputs are folded
    into are the float pass's, bin for bin."""
    from helpers import emul_wide_dft
    rng = np.random.default_rng(100 + R)
    x = rng.normal(size=R) + 1j * rng.normal(size=R)
    ref = np.fft.fft(x)
    assert np.abs(emul_wide_dft(x) - ref).max() < 4e-15 * R * np.abs(ref).max()
    for k in range(R):
        tone = np.exp(2j * np.pi * k * np.arange(R) / R)
        got = emul_wide_dft(tone)
        assert int(np.argmax(np.abs(got))) == k and abs(got[k] - R) < 1e-13 * R
        assert np.abs(np.delete(got, k)).max() < 1e-13 * R
    # the same outputs as the float butterfly, to float precision: the two are interchangeable in a plan
    xf = x.astype(np.complex64)
    assert np.abs(emul_small_dft(xf) - emul_wide_dft(xf.astype(np.complex128))).max() < 3e-7 * np.abs(ref).max()
    # the fused pass the kernels run (inputs converted as they are loaded, floats between the two levels of a composite
    # radix, every output handed on under its natural index): the DFT of the float inputs to one float rounding
    from helpers import emul_wide_pass
    reff = np.fft.fft(xf.astype(np.complex128))
    assert np.abs(emul_wide_pass(xf) - reff).max() < 1.5e-7 * np.abs(reff).max()
    for k in range(R):
        tone = np.exp(2j * np.pi * k * np.arange(R) / R).astype(np.complex64)
        got = emul_wide_pass(tone)
        assert int(np.argmax(np.abs(got))) == k and abs(got[k] - R) < 1e-6 * R and np.abs(np.delete(got, k)).max() < 2e-6 * R


@pytest.mark.parametrize("plan", range(20))
@pytest.mark.parametrize("windowed", [False, True])
def test_emulated_mixed_plan_matches_oracle(plan, windowed):
    """mixed_core.h (the planned mixed-radix kernel's per-thread code: element names, padded slots,
    twiddle indices, packed raw samples, bin placement) for two-, three- and four-pass plans with
    one or several butterflies per thread; plans 13 ... 16 are the split form (N = P x M, P = 2 ... 5), 17 ... 19 its
    paired form (P = 6, 8, 10: sections j and j + P/2 added first)."""
    N = emul_mixed_n(plan)
    R = 12
    stream = rpf.synth.uniform_iq(70 + plan, N * R)
    w = rpf.synth.hann_window(N) + np.float32(0.25) if windowed else None
    got = emul_mixed(plan, stream, R, w)
    assert max_rel(got, truth_f64(N, stream, R, w)) < 1e-6
    o32, _ = oracle_accumulate(N, stream, R, w, 32)
    assert max_rel(got, o32) < 1e-6


def test_emulated_mixed_plan_bin_placement():
    plan = 4                      # 1200 = 10 x 12 x 10
    N = emul_mixed_n(plan)
    k0 = 437
    n = np.arange(N)
    tone = 50.0 * np.exp(2j * np.pi * k0 * n / N)
    frame = np.empty(2 * N, dtype=np.uint8)
    frame[0::2] = np.rint(127 + tone.real)
    frame[1::2] = np.rint(127 + tone.imag)
    got = emul_mixed(plan, frame, 1)
    assert int(np.argmax(got)) == (k0 + N // 2) % N
    assert np.sort(got)[-2] < 1e-3 * got.max()


def test_every_shipped_mixed_radix_plan_on_the_emulator():
    """All plans of mixed_plans.inc and mixed_plans_split.inc -- the tables the library is compiled from, with the
    emulator's thread-by-thread runners in place of the kernels: element names, slots, twiddle indices, bin
    placement (every bin written exactly once) and the split form's residues, against float64 truth; windowed for
    every third size."""
    sizes = emul_shipped_sizes()
    assert len(sizes) > 150 and len(set(sizes)) == len(sizes)
    for i, N in enumerate(sizes):
        R = 2
        stream = rpf.synth.uniform_iq(1000 + N, N * R)
        w = rpf.synth.hann_window(N) + np.float32(0.25) if i % 3 == 0 else None
        got = emul_shipped(i, stream, R, w)
        assert np.all(np.isfinite(got)), N
        # two frames: little averaging, so against max(bin, mean bin) like the other small cases
        assert max_err_over_mean(got, truth_f64(N, stream, R, w)) < 2e-6, N


@pytest.mark.parametrize("n1", [128, 256, 512])
def test_fused_four_step_twiddle_factorisation_holds(n1):
    """rpf_fourstep.hip's fused kernel keeps W_N^{c bin_of(t, a)} as (per-lane register) x (per-register LDS value):
    valid only while bin_of(t, a) = bin_of(t, 0) + bin_of(0, a) for its column geometries."""
    import ctypes
    from helpers import emul_lib
    lib = emul_lib()
    lib.rpf_emul_fused_bin_split.argtypes = [ctypes.c_int]
    assert lib.rpf_emul_fused_bin_split(n1) == 0
    assert lib.rpf_emul_fused_bin_split(64) == -1


@pytest.mark.parametrize("rowb,rows", [(32, 512), (64, 256), (128, 128)])
def test_fused_kernel_raw_row_staging_maps_agree(rowb, rows):
    """csrc/fused_layout.h, the fused four-step kernel's tile of raw rows in LDS: the writer's map (which row and
    piece LDS-DMA lane q fetches) and the reader's (where byte b of row r is) are one bijection, for the three row
    lengths the kernel has (32 bytes x 512 rows ... 128 bytes x 128 rows); and the bank arithmetic behind the
    piece-major choice: the 32 lanes of a ds_read_b32 group read one dword of 32 consecutive rows -- 8 banks piece-major,
    no more than 4 row-major."""
    import ctypes
    from helpers import emul_lib
    lib = emul_lib()
    lib.rpf_emul_fused_raw_stage.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.rpf_emul_fused_raw_bank.argtypes = [ctypes.c_int] * 3
    assert lib.rpf_emul_fused_raw_stage(rowb, rows) == 0
    assert lib.rpf_emul_fused_raw_stage(48, rows) == -1
    banks = {lib.rpf_emul_fused_raw_bank(rowb, row, 4) for row in range(32)}
    assert len(banks) == (8 if rowb > 32 else 4)
