"""The fused four-step kernel (the default of 65536 ... 262144 bins, config C4) is a persistent launch whose 256
workgroups must be co-resident; when they are not -- another process holding a CU -- the launch gives up.  The
reference's worker has no failure path (/root/reference/src/datastore.cxx:48-96): neither may this give-up lose an
acquisition.  Driven here with rpf_debug_fused_fault (csrc/rpf_engine_testing.h: a test hook, not part of the ABI): mode 1 makes a launch give up at once
(a 33rd workgroup on XCD 0), mode 2 is the real thing (a squatter kernel holds one CU until the spins run out).
Run on the GPU box with  pytest -m gpu."""
import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import max_rel, truth_f64
from parity_bars import FUSED_VS_TWO_KERNEL, TRUTH_BAR

pytestmark = pytest.mark.gpu

N = 262144                      # config C4's size; a staging slot holds 64 frames (32 MB)
# These tests are about the protocol -- nothing lost, nothing NaN, the right path afterwards -- on tone streams at C4's
# size, where the asserted accuracy is the distance from float64 TRUTH (parity_bars.py section 4: the CPU path is itself
# ~1.5e-6 off beside the lines there).  `want` below is that truth.
BAR = TRUTH_BAR[N]


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def needs_fused(ds):
    st = ds.fused_status()
    if not st["active"]:
        pytest.skip("this device does not run the fused four-step kernel (not a 256-CU part, or its teams did not assemble)")
    return st


@pytest.fixture(scope="module")
def c4_stream():
    R = 200                     # 4 staging slots: 64 + 64 + 64 + 8 frames
    stream = rpf.synth.noise_tones_iq(4, N * R)
    with rpf.Datastore(rpf.Params(N=N, repeats=R), flags=rpf._lib.FLAG_NO_FOURSTEP_FUSED) as plain:
        assert not plain.fused_status()["active"]
        two_kernel, done = plain.accumulate(stream, R)
    assert done == R
    want = truth_f64(N, stream, R)
    assert max_rel(two_kernel, want) < BAR
    return R, stream, two_kernel, want


def test_fused_abort_falls_back_and_recovers(c4_stream):
    """Every fused launch gives up (the device has become busy for good): rpf_finish returns RPF_OK, the spectrum is
    bit-identical to the two-kernel path's, the engine has left the fused kernel, and the next acquisition on the same
    engine is the two-kernel path's again -- no error, no NaN."""
    R, stream, two_kernel, want = c4_stream
    with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
        needs_fused(ds)
        ok, done = ds.accumulate(stream, R)                     # first: the fused kernel as shipped
        assert done == R and max_rel(ok, want) < BAR
        assert ds.fused_status() == {"active": True, "gave_up": 0, "recovered": 0}
        ds.debug_fused_fault(1, skip=0, count=-1)
        got, done = ds.accumulate(stream, R)                    # rc != 0 would raise
        st = ds.fused_status()
        assert done == R and np.all(np.isfinite(got))
        assert np.array_equal(got, two_kernel)
        assert not st["active"] and st["gave_up"] >= 1 and st["recovered"] == st["gave_up"]
        again, done = ds.accumulate(stream, R)
        assert done == R and np.array_equal(again, two_kernel)
        assert ds.fused_status() == st                          # nothing fused ran any more


@pytest.mark.parametrize("skip", [0, 1, 2, 3])
def test_one_launch_gives_up_in_any_slot(c4_stream, skip):
    """ONE launch of the acquisition gives up -- the first, a middle one (the advisor's case: the next launch clears
    the kernel's own abort flag), the last: the slot's bytes are run again on the two-kernel path, the other slots'
    fused results stay.  Within the bar of float64 truth; and to the all-fused / all-two-kernel spectra within what one slot of
    the other kernel changes."""
    R, stream, two_kernel, want = c4_stream
    with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
        needs_fused(ds)
        ds.debug_fused_fault(1, skip=skip, count=1)
        got, done = ds.accumulate(stream, R)
        st = ds.fused_status()
    assert done == R and np.all(np.isfinite(got))
    assert st["gave_up"] == 1 and st["recovered"] == 1 and not st["active"]
    assert max_rel(got, want) < BAR
    assert max_rel(got, two_kernel) < FUSED_VS_TWO_KERNEL


def test_buffer_protocol_with_straddling_frames_survives_a_give_up():
    """The reference's own hand-off (5 x 1.6 MB buffers, 3.125 frames each: every frame straddles) at N = 262144
    with the second launch giving up: the re-run slot starts with a carried partial frame."""
    R = 150
    stream = rpf.synth.noise_tones_iq(44, N * R + 3000)
    want, wdone = truth_f64(N, stream, R), R
    with rpf.Datastore(rpf.Params(N=N, repeats=R, buf_length=1638400, buffers=5)) as ds:
        needs_fused(ds)
        ds.debug_fused_fault(1, skip=1, count=1)
        ds.begin(R)
        pos = 0
        while pos < stream.size:
            buf = ds.acquire()
            n = min(buf.size, (stream.size - pos) & ~1)
            buf[:n] = stream[pos:pos + n]
            ds.submit(buf, n)
            pos += n
        done = ds.finish()
        st = ds.fused_status()
        assert done == wdone == R
        assert st["gave_up"] == 1 and st["recovered"] == 1
        assert max_rel(ds.pwr, want) < BAR


def test_device_path_give_up_is_loud_and_the_engine_moves_on(torch_dev):
    """rpf_accumulate_device returns without synchronising, so it cannot re-run anything: the spectrum of the launch
    that gave up is NaN (never a wrong number), rpf_fused_status says so once the stream has passed it, and the next
    call runs the two-kernel path."""
    import torch
    R = 40
    stream = rpf.synth.noise_tones_iq(5, N * R)
    want = truth_f64(N, stream, R)
    d_in = torch.from_numpy(stream).to(torch_dev)
    s = torch.cuda.current_stream().cuda_stream
    with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
        needs_fused(ds)
        ds.debug_fused_fault(1, skip=0, count=1)
        d_out = torch.zeros(N, dtype=torch.float64, device=torch_dev)
        assert ds.accumulate_device(d_in.data_ptr(), stream.size, R, d_out.data_ptr(), s) == R
        torch.cuda.synchronize()
        assert torch.isnan(d_out).all()
        st = ds.fused_status()
        assert st == {"active": False, "gave_up": 1, "recovered": 0}
        assert ds.accumulate_device(d_in.data_ptr(), stream.size, R, d_out.data_ptr(), s) == R
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.all(np.isfinite(got)) and max_rel(got, want) < BAR
        assert ds.fused_status() == st


def test_a_cu_held_by_another_kernel_is_survived(c4_stream):
    """The real failure: a kernel of somebody else's holds one CU (here: a squatter wavefront with 96 KB of LDS), the
    256th workgroup cannot start, the others' bounded spins run out -- seconds --, the launch gives up, the squatter
    leaves.  The acquisition still finishes with the right spectrum."""
    R, stream, two_kernel, want = c4_stream
    with rpf.Datastore(rpf.Params(N=N, repeats=R)) as ds:
        needs_fused(ds)
        ds.debug_fused_fault(2, skip=0, count=1)
        got, done = ds.accumulate(stream, R)
        st = ds.fused_status()
    assert done == R and np.all(np.isfinite(got))
    assert st["gave_up"] >= 1 and st["recovered"] == st["gave_up"] and not st["active"]
    assert max_rel(got, want) < BAR


def test_fused_is_what_the_four_step_sizes_run_by_default():
    """(advisor, round 4) the tests of the fused kernel must not pass on the fallback unnoticed: on a 256-CU part the
    default engine of the four-step sizes reports the fused kernel active."""
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("not a 256-CU part")
    for n in (65536, 131072, 262144):
        with rpf.Datastore(rpf.Params(N=n, repeats=8)) as ds:
            assert ds.fused_status()["active"], n
    with rpf.Datastore(rpf.Params(N=4096, repeats=8)) as ds:
        assert ds.fused_status() == {"active": False, "gave_up": 0, "recovered": 0}
