"""The CPU oracle against float64 truth, analytic known answers, the committed
golden vectors and the reference's documented output format.  (-m "not gpu")"""
import ctypes

import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import (C5_CASES, GOLDEN_CASES, OracleWorker, PlanParams, dp, fp, golden_stream, load_golden,
                     max_err_over_mean, max_rel, oracle_accumulate, oracle_lib, truth_f64)

# Tolerances.  float64 oracle vs numpy complex128: pure rounding noise.  float32
# oracle vs truth: accumulated |X|^2 of a float32 FFT, measured 1e-7..3e-7 for
# noise-like input once a few dozen frames are averaged; 5e-7
# leaves the GPU and any FFTW plan room to sit within 1e-6 of each other
# (north_star's parity bar).  With fewer than 16 frames the per-bin relative
# error is ill-conditioned (helpers.max_err_over_mean explains why); those cases
# are held to 1e-6 of the MEAN bin power instead.
TOL64 = 1e-11
TOL32 = 5e-7
TOL32_FEW = 1e-6


def check32(p32, want, R):
    if R >= 16:
        assert max_rel(p32, want) < TOL32
    else:
        assert max_err_over_mean(p32, want) < TOL32_FEW


@pytest.mark.parametrize("N,R", [(2, 5), (8, 9), (30, 5), (64, 20), (500, 7), (512, 100), (1000, 3),
                                 (2018, 3), (4096, 30), (8192, 4)])
@pytest.mark.parametrize("windowed", [False, True])
def test_oracle_matches_float64_truth(N, R, windowed):
    stream = rpf.synth.uniform_iq(100 + N, N * R)
    w = rpf.synth.hann_window(N) + np.float32(0.25) if windowed else None
    t = truth_f64(N, stream, R, w)
    p64, d64 = oracle_accumulate(N, stream, R, w, 64)
    p32, d32 = oracle_accumulate(N, stream, R, w, 32)
    assert d64 == R and d32 == R
    assert max_rel(p64, t) < TOL64
    check32(p32, t, R)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_matches_golden_vectors(name):
    g = load_golden(name)
    N, R = int(g["N"]), int(g["repeats"])
    stream = golden_stream(g)
    w = g.get("window")
    p32, done = oracle_accumulate(N, stream, R, w, 32)
    p64, _ = oracle_accumulate(N, stream, R, w, 64)
    assert done == R
    if "stride" in g:
        st = int(g["stride"])
        assert max_rel(p64[::st], g["pwr"]) < TOL64
        check32(p32[::st], g["pwr"], R)
        assert abs(p32.sum() / float(g["total"]) - 1) < 1e-7
    else:
        assert max_rel(p64, g["pwr"]) < TOL64
        check32(p32, g["pwr"], R)


@pytest.mark.parametrize("name", [C5_CASES[0], C5_CASES[7]])
def test_oracle_matches_c5_hop_fixtures(name):
    """Config C5's hops at full size (5000 frames, seed 50 + hop): first and last hop
    here, all eight on the GPU box."""
    g = load_golden(name)
    N, R = int(g["N"]), int(g["repeats"])
    p32, done = oracle_accumulate(N, golden_stream(g), R, None, 32)
    assert done == R
    assert max_rel(p32, g["pwr"]) < TOL32


def test_known_answer_constant_input():
    # every sample (128,128): x = (1,1)*(-1)^n -> one line at bin N/2 (the DC bin
    # after the (-1)^n centring, datastore.cxx:69-72) of |N (1+i)|^2 per frame
    N, R = 4096, 1000
    stream = np.full(2 * N * R, 128, dtype=np.uint8)
    p, done = oracle_accumulate(N, stream, R)
    assert done == R
    assert p[N // 2] == R * 2.0 * N * N == 33554432000.0      # SURVEY.md 8c tier 3
    other = np.delete(p, N // 2)
    assert np.max(other) <= 1e-9 * p[N // 2]


def test_known_answer_all_127_is_zero():
    N, R = 512, 10
    p, done = oracle_accumulate(N, np.full(2 * N * R, 127, dtype=np.uint8), R)
    assert done == R and np.all(p == 0.0)


@pytest.mark.parametrize("k0", [0, 1, 37, 255, 300])
def test_known_answer_on_bin_tone(k0):
    # x[n] = A exp(+2 pi i k0 n / N) lands in bin (k0 + N/2) mod N
    N, R, A = 512, 4, 60.0
    n = np.arange(N)
    tone = A * np.exp(2j * np.pi * k0 * n / N)
    frame = np.empty(2 * N, dtype=np.uint8)
    frame[0::2] = np.clip(np.rint(127 + tone.real), 0, 255)
    frame[1::2] = np.clip(np.rint(127 + tone.imag), 0, 255)
    p, _ = oracle_accumulate(N, np.tile(frame, R), R)
    assert int(np.argmax(p)) == (k0 + N // 2) % N
    assert p.max() > 0.95 * R * (A * N) ** 2          # quantisation costs a few percent at most


def test_known_answer_impulse_is_flat():
    N, R = 256, 3
    frame = np.full(2 * N, 127, dtype=np.uint8)
    frame[0], frame[1] = 127 + 50, 127 - 20            # x[0] = (50,-20)
    p, _ = oracle_accumulate(N, np.tile(frame, R), R)
    np.testing.assert_allclose(p, R * (50.0 ** 2 + 20.0 ** 2), rtol=1e-6)


def test_window_of_ones_equals_no_window():
    N, R = 1024, 6
    stream = rpf.synth.noise_tones_iq(5, N * R)
    a, _ = oracle_accumulate(N, stream, R, None)
    b, _ = oracle_accumulate(N, stream, R, np.ones(N, dtype=np.float32))
    assert np.array_equal(a, b)


def test_buffers_may_straddle_frames_and_quota_stops_the_worker():
    """datastore.cxx:52,67,81: fft_pointer survives buffer boundaries; frames past
    `repeats` and a trailing partial frame are ignored."""
    N, frames = 64, 23
    stream = rpf.synth.uniform_iq(9, N * frames + 11)            # + a partial frame
    whole, _ = oracle_accumulate(N, stream, frames)
    for sizes in ([2 * N * frames + 22], [10, 50, 128, 2, 1000, 6], [126] * 40, [130] * 40):
        w = OracleWorker(N)
        w.begin(frames)
        pos = 0
        for s in sizes * 3:
            if pos >= stream.size:
                break
            w.consume(stream[pos:pos + s])
            pos += s
        assert w.repeats_done == min(frames, pos // (2 * N))
        if pos >= 2 * N * frames:
            assert np.array_equal(w.pwr, whole)
        w.close()
    # quota smaller than the stream
    w = OracleWorker(N)
    w.begin(5)
    w.consume(stream)
    assert w.repeats_done == 5
    five, _ = oracle_accumulate(N, stream[: 2 * N * 5], 5)
    assert np.array_equal(w.pwr, five)
    # begin() resets everything, including the carried partial frame (datastore.cxx:52)
    w.begin(2)
    w.consume(stream[: 2 * N + 10])
    w.begin(2)
    w.consume(stream[: 2 * N * 2])
    two, _ = oracle_accumulate(N, stream, 2)
    assert np.array_equal(w.pwr, two)
    w.close()


def test_multithreaded_baseline_equals_single_thread():
    N, R = 512, 97
    stream = rpf.synth.noise_tones_iq(12, N * R)
    one, _ = oracle_accumulate(N, stream, R)
    lib = oracle_lib()
    from helpers import u8p
    out = np.zeros(N)
    done = ctypes.c_int64()
    assert lib.rpf_oracle_accumulate_mt(N, None, stream.ctypes.data_as(u8p), stream.size, R, 5,
                                        out.ctypes.data_as(dp), ctypes.byref(done)) == 0
    assert done.value == R and max_rel(out, one) < 1e-13


def test_fft_provider_any_length_against_numpy():
    lib = oracle_lib()
    rng = np.random.default_rng(3)
    for N in (1, 2, 3, 4, 5, 6, 12, 15, 16, 17, 100, 128, 210, 243, 1024, 1009):
        x = rng.standard_normal(2 * N)
        want = np.fft.fft(x[0::2] + 1j * x[1::2])
        plan = lib.rpf_oracle_plan_create(N)
        out64 = np.zeros(2 * N)
        lib.rpf_oracle_fft_f64(plan, x.ctypes.data_as(dp), out64.ctypes.data_as(dp))
        x32 = x.astype(np.float32)
        out32 = np.zeros(2 * N, dtype=np.float32)
        lib.rpf_oracle_fft_f32(plan, x32.ctypes.data_as(fp), out32.ctypes.data_as(fp))
        lib.rpf_oracle_plan_destroy(plan)
        scale = np.max(np.abs(want))
        assert np.max(np.abs((out64[0::2] + 1j * out64[1::2]) - want)) < 1e-13 * scale * max(1, np.log2(N + 1))
        want32 = np.fft.fft(x32[0::2].astype(np.float64) + 1j * x32[1::2].astype(np.float64))
        assert np.max(np.abs((out32[0::2] + 1j * out32[1::2]) - want32)) < 2e-6 * scale


# ---- output stage (acquisition.cxx:360-433) ---------------------------------
def _format(pwr, N, R, freq, sr, linear=0, baseline=None):
    buf = ctypes.create_string_buffer(64 * N + 64)
    p = np.array(pwr, dtype=np.float64)
    b = None if baseline is None else np.ascontiguousarray(baseline, dtype=np.float64).ctypes.data_as(dp)
    n = oracle_lib().rpf_oracle_format_text(p.ctypes.data_as(dp), N, R, freq, sr, linear, b, buf, len(buf))
    assert n > 0
    return buf.value.decode(), p


def test_output_format_matches_the_man_page_example():
    # /root/reference/doc/rtl_power_fftw.1.md:94-99 and :167: -f 1420405752 -b 512 at
    # 2 MS/s prints frequencies 1.41940575e+09, 1.41940966e+09, ... (9 significant
    # digits) and 6-significant-digit powers
    N, R, f, sr = 512, 100, 1420405752, 2000000
    pwr = np.linspace(1e9, 2e9, N)
    text, mutated = _format(pwr, N, R, f, sr)
    lines = text.split("\n")
    assert lines[0].split()[0] == "1.41940575e+09"
    assert lines[1].split()[0] == "1.41940966e+09"
    assert len(lines) == N + 2 and lines[N] == "" and lines[N + 1] == ""   # one blank line after a spectrum
    # DC bin replaced by the mean of its neighbours, in place (acquisition.cxx:377)
    assert mutated[N // 2] == (pwr[N // 2 - 1] + pwr[N // 2 + 1]) / 2
    v = float(lines[7].split()[1])
    want = 10 * np.log10(pwr[7] / R / N / sr)
    assert v == float("%.6g" % want)


def test_output_linear_baseline_and_matrix_row():
    N, R, sr = 64, 10, 2000000
    pwr = np.arange(1, N + 1, dtype=np.float64) * 1e6
    base = np.linspace(-3, 3, N)
    text, _ = _format(pwr, N, R, 100000000, sr, linear=1, baseline=base)
    v = float(text.split("\n")[5].split()[1])
    assert v == float("%.6g" % (pwr[5] / R / N / sr - base[5]))
    row = np.zeros(N, dtype=np.float32)
    p = pwr.copy()
    oracle_lib().rpf_oracle_format_matrix(p.ctypes.data_as(dp), N, R, sr, 0, base.ctypes.data_as(dp),
                                          row.ctypes.data_as(fp))
    p2 = pwr.copy()
    p2[N // 2] = (p2[N // 2 - 1] + p2[N // 2 + 1]) / 2
    np.testing.assert_array_equal(row, (10 * np.log10(p2 / R / N / sr) - base).astype(np.float32))


def test_frequency_precision_rule():
    # significantPlacesFreq = ceil(floor(log10(f)) - log10(sr/N) + 3) with an INTEGER sr/N
    for N, digits in ((512, 9), (4096, 10), (262144, 12)):
        text, _ = _format(np.ones(N), N, 1, 1420405752, 2000000)
        mant = text.split("\n")[1].split()[0].split("e")[0].replace(".", "").lstrip("-")
        assert len(mant) <= digits


# ---- Plan (acquisition.cxx:158-198) ------------------------------------------
def _plan(**kw):
    p = PlanParams(N=512, sample_rate=2000000, repeats=3200, buf_length=1638400, cfreq=1420405752)
    for k, v in kw.items():
        setattr(p, k, v)
    freqs = (ctypes.c_int64 * 64)()
    n = oracle_lib().rpf_oracle_make_plan(ctypes.byref(p), freqs, 64)
    return p, list(freqs[:n])


def test_plan_buffer_lengths_of_the_baseline_configs():
    # C1 -> 114688 B; C2/C5 -> 1638400 B; C4 stays at the default
    p, f = _plan(N=512, repeats=100)
    assert p.buf_length == 114688 and f == [1420405752]
    p, _ = _plan(N=4096, repeats=10000)
    assert p.buf_length == 1638400
    p, _ = _plan(N=262144, repeats=1000)
    assert p.buf_length == 1638400
    p, _ = _plan(N=512, repeats=0)
    assert p.buf_length == 16384


def test_plan_integration_time_and_hops():
    p, _ = _plan(N=512, integration_time_isSet=1, integration_time=10.0)
    assert p.repeats == int(np.ceil(2000000 * 10.0 / 512))
    # -f 100M:116M at 2 MS/s -> 8 hops at 101, 103, ... 115 MHz
    p, f = _plan(N=4096, repeats=5000, freq_hopping_isSet=1, startfreq=100000000, stopfreq=116000000)
    assert f == [101000000 + 2000000 * i for i in range(8)]
    # a span narrower than the bandwidth: one hop in the middle
    p, f = _plan(freq_hopping_isSet=1, startfreq=100000000, stopfreq=101000000)
    assert f == [100500000]
    # overlapping hops cover the range exactly
    p, f = _plan(freq_hopping_isSet=1, startfreq=100000000, stopfreq=105000000)
    assert len(f) == 3 and f[0] == 101000000 and f[-1] + 1000000 == 105000000


def test_producer_read_sizes():
    dn = oracle_lib().rpf_oracle_data_needed
    assert dn(2 * 512 * 100, 0, 114688) == 114688            # C1: 102400 needed -> 7 x 16384
    assert dn(81920000, 0, 1638400) == 1638400
    assert dn(81920000, 81920000 - 1000, 1638400) == 16384
    assert dn(100, 0, 16384) == 16384


def test_fftw_probe_pins_the_oracle_when_the_real_library_exists():
    """The reference's FFT is FFTW3f (datastore.cxx:30-33,82), absent from the build image.
    Where a box has libfftw3f.so.3 the probe (oracle/fftw_probe.py, dlopen only) runs the
    reference's loop around the real fftwf_execute and the oracle must agree with it to the
    parity bar on config C1 and on the head of C2; where it does not, the probe says so."""
    import sys
    sys.path.insert(0, __import__("helpers").ROOT)
    from oracle import fftw_probe
    c1 = rpf.synth.uniform_iq(1, 512 * 100)
    rep = fftw_probe.report(512, c1, 100, {"oracle": oracle_accumulate(512, c1, 100)[0]})
    if fftw_probe.provider() != "fftw3f":
        assert rep["fftw"] == "absent"
        pytest.skip("libfftw3f is not installed here: parity stays unpinned on this box")
    c2 = rpf.synth.noise_tones_iq(2, 4096 * 400)
    rep2 = fftw_probe.report(4096, c2, 400, {"oracle": oracle_accumulate(4096, c2, 400)[0]})
    for r in (rep, rep2):
        assert r["fftw"] == "present" and r["provider"] == "fftw3f"
        for flag in ("measure", "estimate"):
            assert r[flag]["max_rel_vs_oracle"] < 1e-6, r


def test_oracle_against_mkl_through_the_references_own_fftw3_calls():
    """A fifth comparator, through the reference's own call sites: Intel MKL implements the FFTW3
    API (libmkl_rt exports fftwf_plan_dft_1d / fftwf_execute / fftwf_malloc ...), so the probe's restatement of
    datastore.cxx:30-33,66-89 -- plan with FFTW_MEASURE, fill inbuf, fftwf_execute, square and sum in double -- runs around
    it unchanged.  It is MKL's arithmetic, not FFTW's, and pins nothing ("fftw" stays "absent"); what it shows is the
    oracle within the parity bar of a float32 FFT nobody here wrote, called exactly as the reference calls its own:
    C1, the heads of C2 and C3, a non-power-of-two size (the man page's -b 500) and a four-step size."""
    import sys
    sys.path.insert(0, __import__("helpers").ROOT)
    from oracle import fftw_probe
    if fftw_probe.load() is None:
        pytest.skip("neither libfftw3f nor libmkl_rt on this box")
    if fftw_probe.provider() == "fftw3f":
        pytest.skip("this box has the real FFTW: the test above is the one that counts")
    cases = ((512, 100, rpf.synth.uniform_iq(1, 512 * 100), None),
             (4096, 400, rpf.synth.noise_tones_iq(2, 4096 * 400), None),
             (4096, 400, rpf.synth.noise_tones_iq(2, 4096 * 400), rpf.synth.hann_window(4096)),
             (500, 64, rpf.synth.noise_tones_iq(61, 500 * 64), None),
             (65536, 64, rpf.synth.noise_tones_iq(62, 65536 * 64), None))
    for N, R, stream, w in cases:
        orc = oracle_accumulate(N, stream, R, w)[0]
        rep = fftw_probe.report(N, stream, R, {"oracle": orc, "truth": truth_f64(N, stream, R, w)}, w)
        assert rep["fftw"] == "absent" and rep["fftw3_api"]["provider"] == "mkl_fftw3_interface"
        for flag in ("measure", "estimate"):
            fig = rep["fftw3_api"][flag]
            assert fig["frames"] == R
            assert fig["max_rel_vs_oracle"] < 1e-6 and fig["max_rel_vs_truth"] < 1e-6, (N, rep)


def test_oracle_against_mkl_the_third_float32_fft():
    """A float32 FFT of independent provenance that IS on every box: Intel MKL's DFTI through torch.fft on
    CPU tensors (oracle/mkl_probe.py: the reference's loop, datastore.cxx:66-89, around it).  The oracle must
    agree with it to the parity bar on the configurations' streams -- C1, and the first frames of C2 / C3."""
    from oracle import mkl_probe
    if not mkl_probe.available():
        pytest.skip("this torch build has no MKL")
    for N, R, stream, w in ((512, 100, rpf.synth.uniform_iq(1, 512 * 100), None),
                            (4096, 400, rpf.synth.noise_tones_iq(2, 4096 * 400), None),
                            (4096, 400, rpf.synth.noise_tones_iq(2, 4096 * 400), rpf.synth.hann_window(4096))):
        rep = mkl_probe.report(N, stream, R, {"oracle": oracle_accumulate(N, stream, R, w)[0]}, w)
        assert rep["mkl"] == "present" and rep["frames"] == R
        assert rep["max_rel_vs_oracle"] < 1e-6, rep


@pytest.mark.parametrize("N,R,windowed", [(512, 100, False), (4096, 64, False), (4096, 64, True), (500, 64, False),
                                          (66000, 64, False), (100000, 64, False), (262144, 8, False)])
def test_oracle_against_pocketfft_in_float32(N, R, windowed):
    """A fourth float32 FFT of independent provenance, on every box without a GPU or torch: scipy's pocketfft, which
    transforms complex64 input IN single precision (numpy's own fft would promote to double).  The reference's loop
    around it against the oracle, 64 frames of the tone stream: two float32 transforms within the parity bar of each
    other, each within 1e-6 of float64 truth -- and the oracle the closer of the two (measured at 54000 ... 100000 bins:
    pocketfft 0.70 - 0.81e-6 from the truth, the oracle 0.52 - 0.56e-6, the two 0.58 - 0.83e-6 apart: an independent
    float32 FFT says what DESIGN.md 6 says, the bar has no margin left there for anybody).  262144 is a float32-limit size
    (tests/test_gpu_heldout.py): both are ~1.6 - 1.9e-6 from the truth after 8 frames and are only held to 3e-6."""
    from oracle import pocketfft_probe              # the reference's loop around scipy.fft in single precision
    stream = rpf.synth.noise_tones_iq(60 + N % 7, N * R)
    w = rpf.synth.hann_window(N) if windowed else None
    pocket, frames = pocketfft_probe.accumulate(N, stream, R, w)
    assert frames == R
    orc, done = oracle_accumulate(N, stream, R, w)
    assert done == R
    truth = truth_f64(N, stream, R, w)
    bar = 1e-6 if N < 98304 or R >= 64 and N < 131072 else 3e-6
    assert max_rel(orc, pocket) < bar, (N, max_rel(orc, pocket))
    assert max_rel(pocket, truth) < bar and max_rel(orc, truth) < bar
    if N >= 50000 and bar == 1e-6:
        assert max_rel(orc, truth) < max_rel(pocket, truth)
