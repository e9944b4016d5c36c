"""BASELINE.json's full-size configurations on the GPU, checked through
size-independent properties (the oracle would need minutes for these sizes) and
against the oracle on a bounded prefix.  Run with  pytest -m gpu."""
import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import max_rel, oracle_accumulate

pytestmark = pytest.mark.gpu

N, R = 4096, 10000            # config C2 (and C3 with the Hann window)


@pytest.fixture(scope="module")
def c2():
    import torch
    stream = rpf.synth.noise_tones_iq(2, N * R)
    dev = torch.device("cuda:0")
    return stream, torch.from_numpy(stream).to(dev), dev


def device_run(ds, d_in, first_frame, frames, dev):
    import torch
    out = torch.empty(N, dtype=torch.float64, device=dev)
    n = ds.accumulate_device(d_in.data_ptr() + 2 * N * first_frame, 2 * N * frames, frames,
                             out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert n == frames
    return out.cpu().numpy()


@pytest.mark.parametrize("windowed", [False, True])
def test_c2_c3_full_size_properties(c2, windowed):
    stream, d_in, dev = c2
    w = rpf.synth.hann_window(N) if windowed else None
    with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=R), w) as ds:
        full = device_run(ds, d_in, 0, R, dev)
        # (1) additivity over frames: pwr(A ++ B) = pwr(A) + pwr(B), any cut point
        a = device_run(ds, d_in, 0, 3333, dev)
        b = device_run(ds, d_in, 3333, R - 3333, dev)
        assert max_rel(a + b, full) < 1e-12
        # (2) run-to-run reproducibility (deterministic two-stage reduce)
        assert np.array_equal(full, device_run(ds, d_in, 0, R, dev))
        # (3) Parseval: sum_k pwr[k] = N * sum |x[n]|^2, the right side exact in integers
        x = stream.astype(np.int64).reshape(R, N, 2) - 127
        if not windowed:
            energy = float(N * np.sum(x * x))
        else:
            e_n = np.sum(x * x, axis=(0, 2)).astype(np.float64)          # per sample position
            energy = float(N * np.sum(e_n * w.astype(np.float64) ** 2))
        assert abs(full.sum() / energy - 1.0) < 1e-7
        # (4) the oracle on a prefix the CPU finishes in seconds
        head = 400
        want, _ = oracle_accumulate(N, stream[: 2 * N * head], head, w)
        assert max_rel(device_run(ds, d_in, 0, head, dev), want) < 1e-6
        # (5) the queue path over the whole stream (50 reference-sized buffers)
        host, done = ds.accumulate(stream, R)
        assert done == R and max_rel(host, full) < 1e-12


def test_c5_eight_hop_scan_sharded_like_multi_gpu(c2):
    """Config C5's structure on one GPU: 8 hops x 5000 frames, each hop cut into
    the frame ranges two ranks would own; shard sums equal the whole hop."""
    _, d_in, dev = c2
    hops, per_hop = 2, 5000          # two hops' worth of the C2 stream
    with rpf.Datastore(rpf.Params(N=N, repeats=per_hop)) as ds:
        for hop in range(hops):
            whole = device_run(ds, d_in, hop * per_hop, per_hop, dev)
            acc = np.zeros(N)
            for rank in range(2):
                first, count = rpf.sharding.shard_frames(per_hop, 2, rank)
                acc += device_run(ds, d_in, hop * per_hop + first, count, dev)
            assert max_rel(acc, whole) < 1e-12


def test_c4_full_size_properties():
    """Config C4: N = 262144 bins x 1000 repeats (524 MB of IQ, four-step kernels),
    frames straddling the reference's 1.6 MB buffers in the queue path."""
    import torch
    n4, r4 = 262144, 1000
    dev = torch.device("cuda:0")
    stream = rpf.synth.noise_tones_iq(4, n4 * r4)
    d_in = torch.from_numpy(stream).to(dev)

    def run(ds, first, frames):
        out = torch.empty(n4, dtype=torch.float64, device=dev)
        n = ds.accumulate_device(d_in.data_ptr() + 2 * n4 * first, 2 * n4 * frames, frames, out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert n == frames
        return out.cpu().numpy()

    with rpf.Datastore(rpf.Params(N=n4, repeats=r4, buf_length=1638400)) as ds:
        full = run(ds, 0, r4)
        a, b = run(ds, 0, 337), run(ds, 337, r4 - 337)
        assert max_rel(a + b, full) < 1e-12                       # additivity over frames
        assert np.array_equal(full, run(ds, 0, r4))               # reproducible
        x = stream.astype(np.int64).reshape(-1, 2) - 127
        energy = float(n4) * float(np.sum(x * x))
        assert abs(full.sum() / energy - 1.0) < 1e-7              # Parseval, right side exact
        # the oracle on a few frames.  Noise-only bytes: with C4's tones (32768 x the noise
        # floor per bin at this N) any two float32 FFTs -- FFTW plans included -- differ by a
        # few 1e-6 of the floor around them, which says nothing about the kernels.
        head = 6
        noise = rpf.synth.uniform_iq(404, n4 * head)
        d_noise = torch.from_numpy(noise).to(dev)
        out = torch.empty(n4, dtype=torch.float64, device=dev)
        assert ds.accumulate_device(d_noise.data_ptr(), noise.size, head, out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream) == head
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want, _ = oracle_accumulate(n4, noise, head, None)
        floor = np.median(want)
        assert float(np.max(np.abs(got - want) / np.maximum(want, floor))) < 1e-6
        host, done = ds.accumulate(stream, r4)                    # 320 buffers of 3.125 frames
        assert done == r4 and max_rel(host, full) < 1e-12
