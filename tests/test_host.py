"""The C++ host (rtl-power-fftw_amd/host): command line, units, Plan, aux files and
the spectrum writer, against the man page, the oracle's restatements and the
Python mirror.  (-m "not gpu"; the CLI end-to-end test at the bottom is -m gpu)"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import ROOT, PlanParams, dp, fp, oracle_lib

HOST_DIR = os.path.join(ROOT, "rtl-power-fftw_amd", "host")
CLI = os.environ.get("RPF_POWER_CLI") or os.path.join(HOST_DIR, "rpf_power")   # (tools/gpu_tsan.sh: the sanitised build)


@pytest.fixture(scope="module")
def host():
    if not (os.path.exists(os.path.join(HOST_DIR, "librpf_host.so")) and os.path.exists(CLI)):
        subprocess.run(["make", "-C", HOST_DIR], check=True)
    fake_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rtlsdr")
    if not os.path.exists(os.path.join(fake_dir, "libfake_rtlsdr.so")):
        subprocess.run(["make", "-C", fake_dir], check=True)
    import torch  # noqa: F401  (same HIP runtime for librpf_engine.so, see _lib.load)
    lib = ctypes.CDLL(os.path.join(HOST_DIR, "librpf_host.so"))
    lib.rpf_host_parse_frequency.restype = ctypes.c_longlong
    lib.rpf_host_parse_frequency.argtypes = [ctypes.c_char_p]
    lib.rpf_host_parse_time.restype = ctypes.c_double
    lib.rpf_host_parse_time.argtypes = [ctypes.c_char_p]
    lib.rpf_host_next_read_size.restype = ctypes.c_longlong
    lib.rpf_host_next_read_size.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
    lib.rpf_host_format_text.restype = ctypes.c_long
    lib.rpf_host_format_text.argtypes = [dp, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int,
                                         ctypes.c_int, dp, ctypes.c_char_p, ctypes.c_size_t]
    lib.rpf_host_format_matrix.argtypes = [dp, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, dp, fp]
    lib.rpf_host_read_column.argtypes = [ctypes.c_char_p, ctypes.c_int, dp, ctypes.c_int]
    lib.rpf_host_aux_from_stdin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, fp, dp,
                                            ctypes.c_char_p, ctypes.c_size_t]
    lib.rpf_host_synthetic.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong,
                                       ctypes.POINTER(ctypes.c_uint8)]
    return lib


def parse(host, *args):
    argv = (ctypes.c_char_p * (len(args) + 1))(b"rpf_power", *[a.encode() for a in args])
    i = [ctypes.c_int() for _ in range(6)]
    ll = [ctypes.c_longlong() for _ in range(4)]
    d = ctypes.c_double()
    msg = ctypes.create_string_buffer(512)
    rc = host.rpf_host_parse(len(args) + 1, argv, ctypes.byref(i[0]), ctypes.byref(i[1]), ctypes.byref(i[2]),
                             ctypes.byref(ll[0]), ctypes.byref(i[3]), ctypes.byref(ll[1]), ctypes.byref(ll[2]),
                             ctypes.byref(ll[3]), ctypes.byref(i[4]), ctypes.byref(d), msg, len(msg))
    return rc, dict(N=i[0].value, buffers=i[1].value, buf_length=i[2].value, repeats=ll[0].value,
                    sample_rate=i[3].value, cfreq=ll[1].value, startfreq=ll[2].value, stopfreq=ll[3].value,
                    flags=i[4].value, integration_time=d.value), msg.value.decode()


def test_frequency_suffixes(host):
    f = host.rpf_host_parse_frequency
    assert f(b"1420405752") == 1420405752 and f(b"100M") == 100000000 and f(b"1.42G") == 1420000000
    assert f(b"144100k") == 144100000 and f(b"88.5 M") == 88500000
    assert f(b"100X") == -1 and f(b"100MHz") == -1


def test_time_components(host):
    t = host.rpf_host_parse_time
    assert t(b"90") == 90 and t(b"90s") == 90 and t(b"5m") == 300 and t(b"1h30m") == 5400
    assert t(b"2d4h10m5s") == 2 * 86400 + 4 * 3600 + 605 and t(b"1.5h") == 5400 and t(b"1m30") == 90
    assert t(b"5m5m") == -1 and t(b"10x") == -1 and t(b"abc") == -1


def test_defaults_and_derived_values(host):
    rc, o, _ = parse(host)
    # params.h:33-66
    assert rc == 0 and o["N"] == 512 and o["buffers"] == 5 and o["buf_length"] == 1638400
    assert o["repeats"] == 1638400 // 1024 and o["sample_rate"] == 2000000 and o["cfreq"] == 1420405752
    rc, o, _ = parse(host, "-b", "511")                  # odd N is bumped (params.cxx:150-155)
    assert rc == 0 and o["N"] == 512
    rc, o, _ = parse(host, "-s", "100000")               # rounded to x16384 (params.cxx:171-175)
    assert rc == 0 and o["buf_length"] == 98304 and o["flags"] & 256
    rc, o, _ = parse(host, "-b", "4096", "-n", "10000", "-w", "w.txt", "-B", "-", "-l", "-q", "-c")
    assert rc == 0 and o["repeats"] == 10000 and o["flags"] & (1 | 2 | 4 | 8 | 128) == (1 | 2 | 4 | 8 | 128)
    rc, o, _ = parse(host, "--freq", "100M:116M", "--rate=2000000", "--bins", "4096", "-t", "2m")
    assert rc == 0 and o["startfreq"] == 100000000 and o["stopfreq"] == 116000000 and o["cfreq"] == 108000000
    assert o["flags"] & 64 and o["flags"] & 512 and o["integration_time"] == 120


def test_command_line_errors_use_the_reference_exit_codes(host):
    assert parse(host, "-n", "10", "-t", "5")[0] == 3                 # mutually exclusive
    assert parse(host, "-b", "-4")[0] == 3                            # must be positive
    assert parse(host, "-f", "100M:90M")[0] == 3
    assert parse(host, "-f", "12Q")[0] == 3
    assert parse(host, "-t", "bogus")[0] == 3
    assert parse(host, "--nope")[0] == 4                              # TCLAPerror
    assert parse(host, "-b")[0] == 4 and parse(host, "-b", "x")[0] == 4
    assert parse(host, "-b", "8", "-b", "16")[0] == 4
    rc, _, msg = parse(host, "-n", "10", "-t", "5")
    assert msg == "Options -n and -t are mutually exclusive. Exiting."


def plan_via_host(host, samplerate, *args):
    argv = (ctypes.c_char_p * (len(args) + 1))(b"rpf_power", *[a.encode() for a in args])
    rep, bl = ctypes.c_longlong(), ctypes.c_int()
    freqs = (ctypes.c_longlong * 64)()
    host.rpf_host_plan.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    n = host.rpf_host_plan(len(args) + 1, argv, samplerate, ctypes.byref(rep), ctypes.byref(bl), freqs, 64)
    assert n >= 0
    return rep.value, bl.value, list(freqs[:n])


@pytest.mark.parametrize("args,kw", [
    (["-b", "512", "-n", "100"], dict(N=512, repeats=100)),
    (["-b", "4096", "-n", "10000"], dict(N=4096, repeats=10000)),
    (["-b", "262144", "-n", "1000"], dict(N=262144, repeats=1000)),
    (["-b", "512", "-t", "10"], dict(N=512, repeats=1600, integration_time_isSet=1, integration_time=10.0)),
    (["-f", "100M:116M", "-b", "4096", "-n", "5000"], dict(N=4096, repeats=5000, freq_hopping_isSet=1,
                                                          startfreq=100000000, stopfreq=116000000)),
    (["-f", "100M:105M", "-o", "20"], dict(repeats=1600, freq_hopping_isSet=1, startfreq=100000000,
                                           stopfreq=105000000, min_overlap=20.0)),
    (["-f", "144100000:146100000", "-b", "500", "-n", "100"], dict(N=500, repeats=100, freq_hopping_isSet=1,
                                                                  startfreq=144100000, stopfreq=146100000)),
])
def test_plan_equals_the_oracle_restatement(host, args, kw):
    rep, bl, freqs = plan_via_host(host, 2000000, *args)
    p = PlanParams(N=512, sample_rate=2000000, repeats=1600, buf_length=1638400, cfreq=1420405752)
    for k, v in kw.items():
        setattr(p, k, v)
    want = (ctypes.c_int64 * 64)()
    n = oracle_lib().rpf_oracle_make_plan(ctypes.byref(p), want, 64)
    assert (rep, bl, freqs) == (p.repeats, p.buf_length, list(want[:n]))


def test_producer_read_sizes_equal_the_oracle(host):
    for total, done, bl in [(102400, 0, 114688), (81920000, 0, 1638400), (81920000, 81919000, 1638400),
                            (100, 0, 16384), (50000, 16384, 32768), (0, 0, 16384)]:
        assert host.rpf_host_next_read_size(total, done, bl) == oracle_lib().rpf_oracle_data_needed(total, done, bl)


@pytest.mark.parametrize("N,freq,sr,linear,with_base", [(512, 1420405752, 2000000, 0, False),
                                                       (4096, 101000000, 2000000, 0, True),
                                                       (64, 433920000, 250000, 1, True),
                                                       (262144, 1420405752, 2000000, 0, False)])
def test_spectrum_writer_is_byte_identical_to_the_oracle(host, N, freq, sr, linear, with_base):
    rng = np.random.default_rng(N)
    pwr = rng.uniform(1e6, 1e10, N)
    base = rng.uniform(-5, 5, N) if with_base else None
    bptr = None if base is None else base.ctypes.data_as(dp)
    a, b = pwr.copy(), pwr.copy()
    buf_a = ctypes.create_string_buffer(64 * N + 64)
    buf_b = ctypes.create_string_buffer(64 * N + 64)
    na = host.rpf_host_format_text(a.ctypes.data_as(dp), N, 1000, freq, sr, linear, bptr, buf_a, len(buf_a))
    nb = oracle_lib().rpf_oracle_format_text(b.ctypes.data_as(dp), N, 1000, freq, sr, linear, bptr, buf_b, len(buf_b))
    assert na == nb > 0 and buf_a.value == buf_b.value
    assert np.array_equal(a, b)                               # both interpolate the DC bin in place
    ra, rb = np.zeros(N, np.float32), np.zeros(N, np.float32)
    host.rpf_host_format_matrix(pwr.copy().ctypes.data_as(dp), N, 1000, sr, linear, bptr, ra.ctypes.data_as(fp))
    oracle_lib().rpf_oracle_format_matrix(pwr.copy().ctypes.data_as(dp), N, 1000, sr, linear, bptr, rb.ctypes.data_as(fp))
    assert np.array_equal(ra, rb)


def test_man_page_example_first_lines(host):
    # doc/rtl_power_fftw.1.md:94-99: "-f 1420405752 -b 512" starts at 1.41940575e+09, 1.41940966e+09
    N = 512
    pwr = np.full(N, 2.7e9)
    buf = ctypes.create_string_buffer(64 * N)
    host.rpf_host_format_text(pwr.ctypes.data_as(dp), N, 100, 1420405752, 2000000, 0, None, buf, len(buf))
    lines = buf.value.decode().split("\n")
    assert lines[0].startswith("1.41940575e+09 ") and lines[1].startswith("1.41940966e+09 ")
    assert lines[N] == "" and lines[N + 1] == ""


def test_aux_file_grammar(host):
    # doc/rtl_power_fftw.1.md:123-129: last column wins, '#' lines and blank/text lines are skipped
    text = b"# comment\n1.5\n  # indented comment\n100e6 -68.25\n\nnot a number\n1 2 3.25\n7\n"
    out = np.zeros(16)
    n = host.rpf_host_read_column(text, 1, out.ctypes.data_as(dp), 16)
    assert n == 4 and list(out[:4]) == [1.5, -68.25, 3.25, 7.0]
    n = host.rpf_host_read_column(b"0.1\n0.2\n", 0, out.ctypes.data_as(dp), 16)
    assert n == 2 and out[0] == np.float32(0.1) and out[1] == np.float32(0.2)     # window is read as float


def test_aux_both_from_stdin_baseline_first(host):
    N = 4
    text = "\n".join(str(v) for v in [10, 20, 30, 40, 0.1, 0.2, 0.3, 0.4]).encode()
    w, b = np.zeros(N, np.float32), np.zeros(N)
    msg = ctypes.create_string_buffer(256)
    rc = host.rpf_host_aux_from_stdin(N, 1, 1, text, w.ctypes.data_as(fp), b.ctypes.data_as(dp), msg, 256)
    assert rc == 0 and list(b) == [10, 20, 30, 40] and np.allclose(w, [0.1, 0.2, 0.3, 0.4])
    rc = host.rpf_host_aux_from_stdin(N, 1, 1, b"1\n2\n3\n", w.ctypes.data_as(fp), b.ctypes.data_as(dp), msg, 256)
    assert rc == 5 and b"Expected 8 values, found 3" in msg.value          # InvalidInput
    rc = host.rpf_host_aux_from_stdin(N, 1, 0, b"1\n2\n3\n", w.ctypes.data_as(fp), b.ctypes.data_as(dp), msg, 256)
    assert rc == 5 and b"Error reading window function. Expected 4 values, found 3." == msg.value


def test_cpp_synthetic_source_equals_python(host):
    for seed, first, n in [(2, 0, 5000), (52, 12345, 777)]:
        out = np.zeros(2 * n, dtype=np.uint8)
        host.rpf_host_synthetic(seed, first, n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
        want = rpf.synth.noise_tones_iq(seed, first + n)[2 * first:]
        assert np.array_equal(out, want)


def test_cli_exit_codes_without_running_anything(host):
    assert subprocess.run([CLI, "--bogus"], capture_output=True).returncode == 4
    assert subprocess.run([CLI, "-n", "1", "-t", "1"], capture_output=True).returncode == 3
    assert subprocess.run([CLI, "-b", "512", "-w", "/nonexistent/file"], capture_output=True).returncode == 5
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--bins" in r.stdout and "--strict-time" in r.stdout
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([CLI, "-b", "512", "-n", "10", "-q", "--synthetic", "2"], capture_output=True, text=True)
        assert r.returncode == 7 and "no CPU path" in r.stderr              # HardwareError, no fallback


def test_cli_gpus_option_is_validated_before_any_device_is_touched(host):
    assert subprocess.run([CLI, "--gpus", "0,x", "--synthetic", "1"], capture_output=True).returncode == 3
    assert subprocess.run([CLI, "--gpus", "", "--synthetic", "1"], capture_output=True).returncode in (3, 4)
    assert subprocess.run([CLI, "--gpu", "0", "--gpus", "0,1", "--synthetic", "1"], capture_output=True).returncode == 3
    # several devices need a source each device can read on its own: stdin is not one
    r = subprocess.run([CLI, "-b", "512", "-n", "4", "--input", "-", "--gpus", "0,0"], input=b"\0" * 4096,
                       capture_output=True)
    assert r.returncode == 3 and b"needs a source every device can read on its own" in r.stderr


# ------------------------------------------------------------------ live dongle (dlopen of librtlsdr)
FAKE_RTLSDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rtlsdr", "libfake_rtlsdr.so")


def _live_env(tmp_path, **extra):
    env = dict(os.environ, RPF_RTLSDR_LIB=FAKE_RTLSDR, FAKE_RTLSDR_LOG=str(tmp_path / "calls.log"))
    env.update({k: str(v) for k, v in extra.items()})
    return env


def test_cli_without_a_source_wants_a_dongle(host, tmp_path):
    """Neither --input nor --synthetic: the reference's behaviour, a live RTL-SDR
    (device.cxx:29-50) -- and the reference's exit codes when there is none."""
    # no librtlsdr at all (nothing of that name in this image)
    r = subprocess.run([CLI, "-b", "512"], capture_output=True, text=True,
                       env=dict(os.environ, RPF_RTLSDR_LIB="/nonexistent/librtlsdr.so"))
    assert r.returncode == 1 and "No RTL-SDR compatible devices found" in r.stderr
    # library present, no dongle plugged in
    r = subprocess.run([CLI, "-b", "512"], capture_output=True, text=True, env=_live_env(tmp_path, FAKE_RTLSDR_COUNT=0))
    assert r.returncode == 1 and "No RTL-SDR compatible devices found." in r.stderr
    # one dongle, device index 3 asked for
    r = subprocess.run([CLI, "-b", "512", "-d", "3"], capture_output=True, text=True, env=_live_env(tmp_path))
    assert r.returncode == 2 and "Invalid RTL device number. Only 1 devices available." in r.stderr


def test_cli_dongle_setup_sequence(host, tmp_path):
    """rtl_power_fftw.cxx:77-101: gains printed, nearest gain selected in manual
    mode, provisional tuning, ppm, sample rate read back -- recorded by the test
    double.  (On a box without a GPU the run then stops at the engine: exit 7.)"""
    env = _live_env(tmp_path, FAKE_RTLSDR_RATE_OFFSET=-3)
    r = subprocess.run([CLI, "-b", "512", "-n", "10", "-g", "300", "-p", "12", "-f", "1420405752", "-r", "2400000"],
                       capture_output=True, text=True, env=env)
    assert "Available gains (in 1/10th of dB): 0, 9, 14, 27," in r.stderr
    assert "Selected nearest available gain: 297 (29.7 dB)" in r.stderr
    assert "PPM error set to: 12" in r.stderr
    assert "Actual sample rate: 2399997 Hz" in r.stderr
    calls = (tmp_path / "calls.log").read_text().split("\n")
    assert calls[:6] == ["open 0", "gain_mode 1", "gain 297", "freq 1420405752", "ppm 12", "rate 2400000"]
    assert r.returncode in (0, 7)          # 7: no HIP device here


# ------------------------------------------------------------------ GPU: end to end
def _data_lines(text):
    """stdout minus the five '#' header lines (they carry wall-clock timestamps)."""
    return [l for l in text.split("\n") if not l.startswith("#")]


@pytest.mark.gpu
def test_cli_end_to_end_matches_python_mirror(host, tmp_path):
    """rpf_power with a synthetic 8-hop scan: stdout (gnuplot format, one blank line
    between hops, two after the pass) equals the oracle's writer applied to the
    spectra the Python mirror computes from the same bytes."""
    N, R = 512, 100
    args = [CLI, "-f", "100M:104M", "-b", str(N), "-n", str(R), "--synthetic", "7"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hops = [101000000, 103000000]
    want = []
    with rpf.Datastore(rpf.Params(N=N, buf_length=114688, repeats=R)) as ds:
        for f in hops:
            stream = rpf.synth.noise_tones_iq(7 + f % 9973, 114688 // 2)  # one 114688-byte readout, SyntheticSource::seed_for
            pwr, done = ds.accumulate(stream, R)
            assert done == R
            buf = ctypes.create_string_buffer(64 * N)
            oracle_lib().rpf_oracle_format_text(pwr.ctypes.data_as(dp), N, R, f, 2000000, 0, None, buf, len(buf))
            want += buf.value.decode().split("\n")[:-1]
    want += [""]                                                         # second blank line closes the pass
    assert _data_lines(r.stdout)[:-1] == want
    assert r.stdout.count("# rtl-power-fftw output") == 2
    assert "Buffer queue histogram: " in r.stderr and "Actual number of averaged spectra: 100" in r.stderr


@pytest.mark.gpu
def test_cli_file_replay_window_baseline_and_matrix(host, tmp_path):
    N, R = 4096, 40
    stream = rpf.synth.noise_tones_iq(3, N * R)
    iq = tmp_path / "iq.u8"
    iq.write_bytes(stream.tobytes())
    win = rpf.synth.hann_window(N)
    wfile = tmp_path / "hann.txt"
    wfile.write_text("# periodic Hann\n" + "\n".join("%.9g" % v for v in win) + "\n")
    base = np.linspace(-1, 1, N)
    # config C3's working form: window from a file, baseline from stdin
    r = subprocess.run([CLI, "-b", str(N), "-n", str(R), "-f", "1420405752", "--input", str(iq), "-w", str(wfile),
                        "-B", "-", "-q"], input="\n".join("%.17g" % v for v in base) + "\n",
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with rpf.Datastore(rpf.Params(N=N, window=True, repeats=R), win) as ds:
        pwr, done = ds.accumulate(stream, R)
    buf = ctypes.create_string_buffer(64 * N)
    oracle_lib().rpf_oracle_format_text(pwr.ctypes.data_as(dp), N, R, 1420405752, 2000000, 0,
                                        base.ctypes.data_as(dp), buf, len(buf))
    assert _data_lines(r.stdout)[:-1] == buf.value.decode().split("\n")[:-1] + [""]
    # matrix mode: N float32 per scan row + the .met file (doc/rtl_power_fftw.1.md:186-194)
    r = subprocess.run([CLI, "-b", str(N), "-n", str(R), "-f", "1420405752", "--input", str(iq), "-q",
                        "-m", str(tmp_path / "scan")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    row = np.fromfile(tmp_path / "scan.bin", dtype=np.float32)
    assert row.size == N
    met = (tmp_path / "scan.met").read_text().split("\n")
    assert met[0] == "%d # frequency bins (columns)" % N and met[1] == "1 # scans (rows)"
    assert met[4] == "%d # stepFreq (Hz)" % (2000000 // N)


@pytest.mark.gpu
def test_cli_live_dongle_equals_replay_of_the_same_bytes(host, tmp_path):
    """The live path (librtlsdr through dlopen, here the file-backed test double)
    and --input replay of the same bytes print the same spectra."""
    N, R = 1024, 64
    stream = rpf.synth.noise_tones_iq(11, N * R + 4096)
    iq = tmp_path / "iq.u8"
    iq.write_bytes(stream.tobytes())
    common = ["-b", str(N), "-n", str(R), "-f", "433920000", "-q"]
    live = subprocess.run([CLI] + common, capture_output=True, text=True, env=_live_env(tmp_path, FAKE_RTLSDR_FILE=iq))
    replay = subprocess.run([CLI] + common + ["--input", str(iq)], capture_output=True, text=True)
    assert live.returncode == 0 and replay.returncode == 0, live.stderr + replay.stderr
    assert _data_lines(live.stdout) == _data_lines(replay.stdout)
    assert len(_data_lines(live.stdout)) >= N
    calls = (tmp_path / "calls.log").read_text().split("\n")
    assert "freq 433920000" in calls and calls[-2] == "close"


@pytest.mark.gpu
def test_cli_replay_file_that_ends_on_the_last_frame(host, tmp_path):
    """A replay file holding exactly 2*N*repeats bytes that is not a multiple of 16384: the
    producer's last request is rounded up to whole USB transfers (acquisition.cxx:288-300),
    the read comes back short, and the bytes it did deliver are the tail of the data."""
    for N, R in ((500, 20), (512, 100), (4096, 3)):
        stream = rpf.synth.uniform_iq(40 + N, N * R)
        assert stream.size % 16384 != 0
        iq = tmp_path / ("iq_%d.u8" % N)
        iq.write_bytes(stream.tobytes())
        r = subprocess.run([CLI, "-b", str(N), "-n", str(R), "-f", "1420405752", "--input", str(iq)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Actual number of averaged spectra: %d" % R in r.stderr and "nan" not in r.stdout.lower()
        with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
            pwr, done = ds.accumulate(stream, R)
        assert done == R
        buf = ctypes.create_string_buffer(64 * N)
        oracle_lib().rpf_oracle_format_text(pwr.ctypes.data_as(dp), N, R, 1420405752, 2000000, 0, None, buf, len(buf))
        assert _data_lines(r.stdout)[:-1] == buf.value.decode().split("\n")[:-1] + [""]
    # a file with no whole frame at all: no NaN spectrum, AcquisitionError
    iq = tmp_path / "short.u8"
    iq.write_bytes(bytes(100))
    r = subprocess.run([CLI, "-b", "512", "-n", "10", "--input", str(iq)], capture_output=True, text=True)
    assert r.returncode == 6 and "nan" not in r.stdout.lower()
    # --continue on a finite replay ends with the data instead of spinning
    iq = tmp_path / "two.u8"
    iq.write_bytes(rpf.synth.uniform_iq(9, 512 * 20).tobytes())
    r = subprocess.run([CLI, "-b", "512", "-n", "10", "--input", str(iq), "-c", "-q"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.count("# rtl-power-fftw output") == 2


@pytest.mark.gpu
def test_cli_scan_spread_over_several_engines(host, tmp_path):
    """--gpus a,b,...: one engine per listed device, hops dealt hop-major, spectra written in
    hop order.  Two engines on device 0 print byte for byte what one engine
    prints for the 8-hop scan of config C5's shape (whole hops per engine: the very same
    sums); three engines cut hops at frame boundaries and their partial sums are added in
    device order -- equal to the last printed digit."""
    N, R = 4096, 500
    scan = ["-f", "100M:116M", "-r", "2000000", "-b", str(N), "-n", str(R), "--synthetic", "50", "-q"]
    one = subprocess.run([CLI] + scan + ["--gpus", "0"], capture_output=True, text=True)
    two = subprocess.run([CLI] + scan + ["--gpus", "0,0"], capture_output=True, text=True)
    three = subprocess.run([CLI] + scan + ["--gpus", "0,0,0"], capture_output=True, text=True)
    assert one.returncode == 0 and two.returncode == 0 and three.returncode == 0, one.stderr + two.stderr + three.stderr
    assert one.stdout.count("# rtl-power-fftw output") == 8
    assert _data_lines(two.stdout) == _data_lines(one.stdout)
    a, b = _data_lines(one.stdout), _data_lines(three.stdout)
    assert len(a) == len(b)
    for la, lb in zip(a, b):
        if la != lb:                                 # a last-digit difference of the 6-digit dB value
            fa, fb = la.split(), lb.split()
            assert fa[0] == fb[0] and abs(float(fa[1]) - float(fb[1])) <= 2e-5 * abs(float(fa[1]))
    # where the per-device spectra are added: RCCL refuses one device listed twice, so these runs fell back to the
    # host (that is the default's contract); asking for RCCL explicitly must then fail loudly, --reduce host must not
    forced = subprocess.run([CLI] + scan + ["--gpus", "0,0", "--reduce", "rccl"], capture_output=True, text=True)
    assert forced.returncode == 7 and "--reduce rccl" in forced.stderr, forced.stderr
    on_host = subprocess.run([CLI] + scan + ["--gpus", "0,0", "--reduce", "host"], capture_output=True, text=True)
    assert on_host.returncode == 0 and _data_lines(on_host.stdout) == _data_lines(one.stdout)
    # a single hop over two engines: frame ranges of the one hop, summed
    single = ["-f", "433920000", "-b", "1024", "-n", "300", "--synthetic", "3", "-q"]
    s1 = subprocess.run([CLI] + single, capture_output=True, text=True)
    s2 = subprocess.run([CLI] + single + ["--gpus", "0,0"], capture_output=True, text=True)
    assert s1.returncode == 0 and s2.returncode == 0, s1.stderr + s2.stderr
    for la, lb in zip(_data_lines(s1.stdout), _data_lines(s2.stdout)):
        if la != lb:
            fa, fb = la.split(), lb.split()
            assert fa[0] == fb[0] and abs(float(fa[1]) - float(fb[1])) <= 2e-5 * abs(float(fa[1]))
    # file replay: every engine seeks to its own hops of the one file; matrix rows in hop order
    hops, n, r = 4, 512, 64
    stream = rpf.synth.noise_tones_iq(77, hops * 65536 // 2)          # 4 hops x one 65536-byte readout
    iq = tmp_path / "scan.u8"
    iq.write_bytes(stream.tobytes())
    args = ["-f", "100M:108M", "-b", str(n), "-n", str(r), "--input", str(iq), "-q"]
    f1 = subprocess.run([CLI] + args, capture_output=True, text=True)
    f2 = subprocess.run([CLI] + args + ["--gpus", "0,0"], capture_output=True, text=True)
    assert f1.returncode == 0 and f2.returncode == 0, f1.stderr + f2.stderr
    assert f1.stdout.count("# rtl-power-fftw output") == hops and _data_lines(f1.stdout) == _data_lines(f2.stdout)
    m1 = subprocess.run([CLI] + args + ["-m", str(tmp_path / "m1")], capture_output=True, text=True)
    m2 = subprocess.run([CLI] + args + ["-m", str(tmp_path / "m2"), "--gpus", "0,0"], capture_output=True, text=True)
    assert m1.returncode == 0 and m2.returncode == 0, m1.stderr + m2.stderr
    assert (tmp_path / "m1.bin").read_bytes() == (tmp_path / "m2.bin").read_bytes()
    assert (tmp_path / "m1.met").read_text().split("\n")[:5] == (tmp_path / "m2.met").read_text().split("\n")[:5]


def test_mixed_plan_generator_emits_valid_candidates():
    """tools/gen_mixed_plans.py (the candidate plans the GPU search times): every candidate of every size multiplies
    to its length, stays within 1024 threads and 160 KB of LDS, and the numbering the picker relies on is stable."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_mixed_plans", os.path.join(ROOT, "tools", "gen_mixed_plans.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    total = 0
    for n in gen.SIZES[::7] + [500, 7000, 10000]:
        cands = gen.candidates(n)
        assert cands == gen.candidates(n)
        for cost, rad, gs, fpw, tw in cands:
            prod = 1
            for r, g in zip(rad, gs):
                prod *= r
                assert r in gen.RADICES and n % (r * g) == 0 and r * g <= 32
            assert prod == n and tw in (0, 1, 2)
            assert fpw * max(n // (r * g) for r, g in zip(rad, gs)) <= 1024
            assert gen.lds_bytes(n, rad, gs, fpw, tw) <= 163840
            assert "MixPlan<%d, %d, %d," % (n, fpw, tw) in gen.entry(n, rad, gs, fpw, tw, 1)
        total += len(cands)
    assert total > 100
