"""Held-out parity of every size whose kernel plan was PICKED.

tools/gpu_parity_score.py and the plan pickers score split-form candidates on the seeds 300 + N % 89 and
1300 + N % 97, and test_gpu_parity.test_tone_stream_parity_where_the_margin_is_thin asserts on those very streams -- a
plan chosen for its error on a stream passes on that stream by construction.  This file runs the same 64-frame
noise + tones measurement on streams no picker has seen.

Two generations of streams, and they are NOT the same kind of evidence:

  * held_out_a / _b (round 4) and _c (round 5) WERE held out when they were introduced; since round 5 they are
    TUNING streams: tools/analysis/parity_passes.py reads their seeds, and mixed_core.h's choice of which split forms
    run their last pass(es) in double was made on "the worst held-out stream".  They stay as regression streams (marked
    `slow`: tools/gpu_r06.sh runs them, the driver's `pytest -m gpu` does not).
  * held_out_d / _e (round 6): fresh keys that live in THIS file only.  No tool under tools/ reads them, scores on
    them or may ever do so (tests/test_plan_tools.py greps tools/ for the key constants).  held_out_d runs in the
    driver's `pytest -m gpu`; held_out_e is `slow`.

The rule, written before round 6's measurement: a case holds if the GPU is within PARITY of the CPU path.  A case
that does not hold is a CONTRACT DEVIATION and is tolerated only if, on that stream, the CPU path is itself
>= parity_bars.ORACLE_OFF from float64 truth while the GPU is within parity_bars.VS_TRUTH of it
(parity_bars.deviation_is_justified); it is then listed BY NAME in parity_bars.CONTRACT_DEVIATIONS and in README.md.
Anything else that fails here leaves mixed_plans_split.inc (large Bluestein takes the size back); it is not re-picked
on these seeds.  The sizes of parity_bars.TRUTH_BAR (98304 bins and up on these streams: the CPU path itself is
0.9 - 2.4e-6 from the truth) are asserted against float64 truth; their distance from the CPU path is recorded.

History.  Round 4: ten split-form sizes failed on a / b and left the table.  Round 5: parity_passes found where the
split / paired forms lose their accuracy (the LAST pass, beside a strong line), the forms of 40000 bins and more run
that pass in double, and the ten sizes came back on the plans they had.  Round 6: the four-step kernels run the row
transform's last pass in double (rpf_fourstep.hip, fourstep_is_wide) and, from 131072 bins up, the pass before it with
twiddles exact to double precision (fourstep_wide2).

The errors are written to the file $RPF_PARITY_RECORD names (tools/gpu_r06.sh sets it ->
profiles/rNN_fullsize_errors.json), else to pytest's tmp_path: running the tests has no side effect on the tree."""
import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import max_err_over_mean, max_rel, oracle_accumulate, truth_f64
from parity_bars import TRUTH_BAR, VS_TRUTH
from test_gpu_parity import (THIN_MARGIN_SIZES, assert_tone_stream_bars, record_errors, run_device,  # noqa: F401
                             tone_stream_errors, torch_dev)

pytestmark = pytest.mark.gpu

# The keys.  Do not use them, or the formulas below, anywhere under tools/.
HELD_OUT_KEY = 0x52304F34_48454C44          # rounds 4 - 5: streams a, b (tuning streams since round 5)
HELD_OUT_KEY_C = 0x52303543_54484952        # round 5: stream c (likewise)
HELD_OUT_KEY_R6 = 0x52303644_5F465245       # round 6: streams d, e -- held out


def _mix(x):
    """splitmix64's finaliser: decorrelates the seeds of neighbouring sizes."""
    x &= 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def held_out_seed(name, N):
    if name == "held_out_a":
        return (HELD_OUT_KEY ^ (N * 0x9E3779B97F4A7C15)) & 0x7FFFFFFF
    if name == "held_out_b":
        return ((HELD_OUT_KEY >> 17) + 7919 * N * N + 104729) & 0x7FFFFFFF
    if name == "held_out_c":
        return ((HELD_OUT_KEY_C * (2 * N + 1)) ^ (HELD_OUT_KEY_C >> 29) ^ (N << 7)) & 0x7FFFFFFF
    if name == "held_out_d":
        return _mix(HELD_OUT_KEY_R6 + 2 * N) & 0x7FFFFFFF
    if name == "held_out_e":
        return _mix((HELD_OUT_KEY_R6 >> 3) ^ (N * 0xD1B54A32D192ED03) ^ 0x65) & 0x7FFFFFFF
    raise KeyError(name)


def tuning_stream_seeds(N):
    """Streams a, b, c -- the ONLY seeds a tool under tools/ may import (gpu_heldout_alternatives.py,
    analysis/parity_passes.py: they measure, they pick nothing).  test_plan_tools.py checks that no tool names d or e."""
    return [(name, held_out_seed(name, N)) for name in ("held_out_a", "held_out_b", "held_out_c")]


slow = pytest.mark.slow
STREAMS = ["held_out_d", pytest.param("held_out_e", marks=slow), pytest.param("held_out_a", marks=slow),
           pytest.param("held_out_b", marks=slow), pytest.param("held_out_c", marks=slow)]

# every size a picker chose a plan for, in any round, and the four-step sizes; the sizes of TRUTH_BAR have the test below
PICKED_SIZES = sorted(n for n in set(THIN_MARGIN_SIZES) if n not in TRUTH_BAR)


@pytest.mark.parametrize("stream", STREAMS)
@pytest.mark.parametrize("N", PICKED_SIZES)
def test_picked_sizes_hold_the_bar_on_streams_no_picker_has_seen(N, stream, torch_dev, tmp_path):
    """64 frames of the noise + tones stream (deterministic lines 1e4 above the weakest bins: a float32 FFT's
    rounding error is coherent there and does not average down), rectangular and Hann, at every split-form /
    paired-form size and 16384 / 32768 / 65536: GPU against the CPU path at PARITY, named deviations excepted
    (parity_bars.py section 5); pocketfft in single precision recorded beside it as a second comparator."""
    out = tone_stream_errors(N, held_out_seed(stream, N), torch_dev, second_comparator=True)
    record_errors(tmp_path, stream, N, out)
    assert_tone_stream_bars(N, stream, out)


@pytest.mark.parametrize("stream", STREAMS)
@pytest.mark.parametrize("N", sorted(TRUTH_BAR))
def test_where_float32_gives_out(N, stream, torch_dev, tmp_path):
    """The sizes of parity_bars.TRUTH_BAR on the held-out tone streams, rectangular and Hann (524288: one stream,
    rectangular).  The CPU path itself is 0.9 - 2.4e-6 from float64 truth in its worst bin there, so what is asserted of the
    GPU is its distance from the TRUTH, in every bin: TRUTH_BAR[N].  Its distance from the CPU path is what it is between
    a transform near the truth and one that is not -- recorded, not asserted; so is that distance over the bins where the
    CPU path is itself within VS_TRUTH of the truth (>= 99.9 % of them: asserted, it says the comparison is meaningful)."""
    if N > 262144 and stream != "held_out_d":
        pytest.skip("524288 bins (catch-all path): one stream")
    seed = held_out_seed(stream, N)
    data = rpf.synth.noise_tones_iq(seed, N * 64)
    for windowed in ((False,) if N > 262144 else (False, True)):
        w = rpf.synth.hann_window(N) if windowed else None
        with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=64), w) as ds:
            got, n = run_device(ds, data, 64, torch_dev)
        assert n == 64
        o32, _ = oracle_accumulate(N, data, 64, w, 32)
        truth = truth_f64(N, data, 64, w)
        good = np.abs(o32 - truth) < VS_TRUTH * truth
        e = {"gpu_vs_oracle": max_rel(got, o32), "gpu_vs_truth": max_rel(got, truth), "oracle_vs_truth": max_rel(o32, truth),
             "gpu_vs_oracle_over_mean": max_err_over_mean(got, o32),
             "bins_where_cpu_is_within_5e-7_of_truth": float(good.mean()),
             "gpu_vs_oracle_on_those_bins": float(np.max(np.abs(got[good] - o32[good]) / o32[good]))}
        record_errors(tmp_path, stream + "_float32_limit", N, {"hann" if windowed else "rect": e})
        assert e["gpu_vs_truth"] < TRUTH_BAR[N], (N, stream, windowed, e)
        assert e["bins_where_cpu_is_within_5e-7_of_truth"] > 0.999, (N, stream, windowed, e)
