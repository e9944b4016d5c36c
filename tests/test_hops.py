"""The multi-hop launch (rtl-power-fftw_amd/csrc/hop_partition.h): how one persistent launch of
the fused kernel walks the hops of a scan.  CPU part: the host-side partition and the kernel's
cursor walk, restated in tests/emul around the very same HopCursor / partition_hops; GPU part
(-m gpu): rpf_accumulate_device_hops against the oracle, the per-hop entry and the C5 fixtures."""
import ctypes

import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import (C5_CASES, emul_lib, golden_stream_device, load_golden, max_err_over_mean, max_rel,
                     oracle_accumulate)

MAX_HOPS = 16


def walk(nframes, fpw, max_grid, rawd=2, interleave=True):
    lib = emul_lib()
    H = len(nframes)
    nf = (ctypes.c_int64 * H)(*nframes)
    it_begin = (ctypes.c_int * (H + 1))()
    slot_begin = (ctypes.c_int * (H + 1))()
    visits = [np.zeros(max(1, n), np.int32) for n in nframes]
    vptrs = (ctypes.POINTER(ctypes.c_int32) * H)(*[v.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) for v in visits])
    cap = max_grid + MAX_HOPS
    slot_hop = (ctypes.c_int * cap)()
    staged = ctypes.c_long()
    lib.rpf_emul_walk_hops.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_long)]
    grid = lib.rpf_emul_walk_hops(nf, H, fpw, max_grid, rawd, 1 if interleave else 0, it_begin, slot_begin, vptrs, slot_hop, cap,
                                  ctypes.byref(staged))
    return grid, list(it_begin), list(slot_begin), visits, list(slot_hop), staged.value


CASES = [
    ([10000], 2, 256),                                  # C2: a single acquisition
    ([5000] * 8, 2, 256),                               # C5: the 8-hop scan
    ([5000] * 16, 2, 256),
    ([1, 0, 3, 0, 0, 7, 2], 2, 256),                    # fewer iterations than workgroups, empty hops
    ([0, 0, 0], 4, 64),                                 # nothing at all
    ([513, 1, 1, 1, 10000, 2, 777], 4, 304),
    ([100] * 16, 32, 1024),                             # N = 64: 32 frames per workgroup
    ([3, 5, 100000], 1, 512),
]


@pytest.mark.parametrize("nframes,fpw,max_grid", CASES)
@pytest.mark.parametrize("rawd", [1, 2, 4])
@pytest.mark.parametrize("interleave", [True, False])
def test_every_frame_is_visited_once_and_every_slot_written_once(nframes, fpw, max_grid, rawd, interleave):
    grid, it_begin, slot_begin, visits, slot_hop, staged = walk(nframes, fpw, max_grid, rawd, interleave)
    H = len(nframes)
    total = sum(-(-n // fpw) for n in nframes)
    assert grid == min(max_grid, total)
    assert it_begin[0] == 0 and it_begin[H] == total
    for h, n in enumerate(nframes):
        assert it_begin[h + 1] - it_begin[h] == -(-n // fpw)
        if n:
            assert np.all(visits[h][:n] == 1), "hop %d: frames missed or visited twice" % h
    # slot ranges: contiguous, in hop order, each slot written exactly once by its hop; hops
    # without a frame own no slot (the reduce then writes zeros)
    assert slot_begin[0] == 0
    for h, n in enumerate(nframes):
        rng = range(slot_begin[h], slot_begin[h + 1])
        assert (len(rng) == 0) == (n == 0)
        assert all(slot_hop[s] == h for s in rng)
        assert len(rng) <= grid
    assert slot_begin[H] <= grid + H - 1 if total else slot_begin[H] == 0
    assert all(s == -1 for s in slot_hop[slot_begin[H]:])
    # the DMA ring issues the same number of stagings per iteration, `rawd` more per workgroup
    assert staged == total + rawd * grid


def test_partition_rejects_what_does_not_fit():
    grid, *_ = walk([1] * 17, 2, 256)
    assert grid == -1
    grid, *_ = walk([-1], 2, 256)
    assert grid == -1


# ------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def run_hops(ds, streams, repeats, torch_dev, split=False):
    import torch
    N = ds.params.N
    d_in = [torch.from_numpy(np.ascontiguousarray(s)).to(torch_dev) if len(s) else
            torch.zeros(16, dtype=torch.uint8, device=torch_dev) for s in streams]
    d_out = torch.full((len(streams), N), float("nan"), dtype=torch.float64, device=torch_dev)
    s = torch.cuda.current_stream().cuda_stream
    ptrs = [d.data_ptr() for d in d_in]
    nbytes = [len(x) for x in streams]
    if split:
        done = ds.device_fused_hops(ptrs, nbytes, repeats, s)
        ds.device_reduce(d_out.data_ptr(), s)
    else:
        done = ds.accumulate_device_hops(ptrs, nbytes, repeats, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), done


@pytest.mark.gpu
@pytest.mark.parametrize("N", [64, 512, 1024, 4096, 8192])
@pytest.mark.parametrize("windowed", [False, True])
def test_hops_entry_matches_oracle_and_single_entry(N, windowed, torch_dev):
    """Ragged hops -- empty ones, one shorter than a workgroup's frame slots, a quota below the
    stream, a trailing partial frame -- in ONE launch: every hop equals the CPU path and the
    one-acquisition entry (same frames, another grouping of the f64 partial sums)."""
    import torch
    frames = [37, 0, 1, 300 * 4096 // N, 5, 0, 64]
    rng = np.random.default_rng(N)
    streams = [rpf.synth.noise_tones_iq(100 + N + h, N * f) for h, f in enumerate(frames)]
    streams[4] = np.concatenate([streams[4], rng.integers(0, 256, 2 * (N // 2), dtype=np.uint8)])   # half a frame more
    repeats = [f for f in frames]
    repeats[0] = 30                          # quota below what the stream holds
    repeats[6] = 1000                        # quota above it
    w = rpf.synth.hann_window(N) + np.float32(0.25) if windowed else None
    with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=1), w) as ds:
        got, done = run_hops(ds, streams, repeats, torch_dev)
        got2, done2 = run_hops(ds, streams, repeats, torch_dev, split=True)
        assert done == done2 == [30, 0, 1, frames[3], 5, 0, 64]
        assert np.array_equal(got, got2)
        s = torch.cuda.current_stream().cuda_stream
        for h, st in enumerate(streams):
            want, wdone = oracle_accumulate(N, st, repeats[h], w, 32)
            assert wdone == done[h]
            if wdone == 0:
                assert np.all(got[h] == 0.0)
                continue
            # (per-bin relative error of a few-frame average is ill-conditioned: helpers.max_err_over_mean)
            # (a one-frame 'average' of a line spectrum sits at the float32 limit of any FFT; the exact check
            # for those hops is the single-acquisition entry below)
            assert (max_rel(got[h], want) < 1e-6) if wdone >= 16 else (max_err_over_mean(got[h], want) < 5e-6), (h, wdone)
            d_in = torch.from_numpy(st).to(torch_dev)
            d_one = torch.empty(N, dtype=torch.float64, device=torch_dev)
            ds.accumulate_device(d_in.data_ptr(), st.size, repeats[h], d_one.data_ptr(), s)
            torch.cuda.synchronize()
            assert max_rel(got[h], d_one.cpu().numpy()) < 1e-12


@pytest.mark.gpu
def test_hops_entry_beyond_one_launch_and_repeatable(torch_dev):
    """More hops than one launch takes (two launches + remainder), bit-identical when repeated."""
    N, H = 2048, 37
    frames = [3 + (7 * h) % 23 for h in range(H)]
    streams = [rpf.synth.uniform_iq(900 + h, N * f) for h, f in enumerate(frames)]
    with rpf.Datastore(rpf.Params(N=N, repeats=1)) as ds:
        assert ds.max_hops_per_launch() == MAX_HOPS
        a, done = run_hops(ds, streams, frames, torch_dev)
        b, _ = run_hops(ds, streams, frames, torch_dev)
        assert done == frames and np.array_equal(a, b)
        for h in (0, 15, 16, 31, 32, 36):
            want, _ = oracle_accumulate(N, streams[h], frames[h], None, 32)
            assert max_err_over_mean(a[h], want) < 1e-6
        with pytest.raises(rpf.RPFError):
            run_hops(ds, streams, frames, torch_dev, split=True)     # the split form is one launch only


@pytest.mark.gpu
@pytest.mark.parametrize("N", [500, 16384, 3000])
def test_hops_entry_on_the_other_kernel_families(N, torch_dev):
    """Sizes outside the LDS-resident kernel run hop by hop behind the same entry."""
    frames = [9, 0, 20]
    streams = [rpf.synth.uniform_iq(N + h, N * f) for h, f in enumerate(frames)]
    with rpf.Datastore(rpf.Params(N=N, repeats=1)) as ds:
        got, done = run_hops(ds, streams, frames, torch_dev)
        assert done == frames and np.all(got[1] == 0.0)
        for h in (0, 2):
            want, _ = oracle_accumulate(N, streams[h], frames[h], None, 32)
            assert max_err_over_mean(got[h], want) < 1e-6
        with pytest.raises(rpf.RPFError):
            run_hops(ds, streams, frames, torch_dev, split=True)


@pytest.mark.gpu
def test_c5_scan_in_one_launch_matches_the_fixtures(torch_dev):
    """BASELINE.json configs[4] at full size through the hops entry: 8 hops x 5000 frames of
    N = 4096 (seeds 50..57) in one persistent launch + one reduce, against the committed
    float64 fixtures (north_star's bar, per bin, plain max-rel)."""
    import torch
    gs = [load_golden(c) for c in C5_CASES]
    N, R = int(gs[0]["N"]), int(gs[0]["repeats"])
    d_in = [golden_stream_device(g, torch_dev)[1] for g in gs]
    d_out = torch.empty((8, N), dtype=torch.float64, device=torch_dev)
    with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
        done = ds.accumulate_device_hops([d.data_ptr() for d in d_in], [d.numel() for d in d_in], [R] * 8,
                                         d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    assert done == [R] * 8
    got = d_out.cpu().numpy()
    worst = max(max_rel(got[h], gs[h]["pwr"]) for h in range(8))
    assert worst < 1e-6, worst


@pytest.mark.gpu
def test_scan_reducer_over_rccl_with_one_device(torch_dev):
    """rpf_scan_reducer_* (the product's final reduce) with the one device this box has:
    RCCL loads (dlopen), ncclCommInitAll / ncclReduce run with a single rank, and the block that comes
    back holds exactly what the engines deposited -- rows of hops nobody deposited stay zero."""
    N, R, H = 4096, 40, 3
    lib = rpf.load()
    devs = (ctypes.c_int * 1)(0)
    red = ctypes.c_void_p()
    rc = lib.rpf_scan_reducer_create(devs, 1, N, H, ctypes.byref(red))
    assert rc == 0, lib.rpf_scan_reducer_last_error(None)
    try:
        want = {}
        assert lib.rpf_scan_reducer_begin(red) == 0
        with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
            for hop in (2, 0):                                       # hop 1 is "another device's"
                pwr, done = ds.accumulate(rpf.synth.noise_tones_iq(60 + hop, N * R), R)
                assert done == R
                want[hop] = pwr
                assert lib.rpf_scan_reducer_deposit(red, 0, hop, ds._handle) == 0
        out = np.full((H, N), np.nan)
        assert lib.rpf_scan_reducer_reduce(red, H, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
        assert np.array_equal(out[0], want[0]) and np.array_equal(out[2], want[2]) and np.all(out[1] == 0.0)
        # argument errors are reported, not executed
        assert lib.rpf_scan_reducer_deposit(red, 1, 0, ds._handle) == 3
        assert lib.rpf_scan_reducer_reduce(red, H + 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 3
    finally:
        lib.rpf_scan_reducer_destroy(red)
    # a device listed twice is something RCCL refuses: create fails with the hardware code, nothing leaks
    two = (ctypes.c_int * 2)(0, 0)
    rc = lib.rpf_scan_reducer_create(two, 2, N, H, ctypes.byref(red))
    assert rc == 7 and not red.value
