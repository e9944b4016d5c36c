"""Held-out parity of every size whose kernel plan was PICKED.

tools/gpu_parity_score.py and the plan pickers score split-form candidates on the seeds 300 + N % 89 and
1300 + N % 97, and test_tone_stream_parity_where_the_margin_is_thin asserts on those very streams -- a plan chosen
for its error on a stream passes on that stream by construction.  The seeds below come from a constant that lives
in THIS file only; no tool under tools/ reads it, scores on it or has ever seen the streams it names.  A size that
fails here is not re-picked on these seeds: it leaves mixed_plans_split.inc (large Bluestein takes it back) or gets
a plan that is more accurate by construction, and DESIGN.md 6 lists it.

Bar: BASELINE.json north_star, "<= 1e-6 per-bin relative error" against the CPU path (oracle/rpf_oracle.c,
a restatement of /root/reference/src/datastore.cxx:66-89), plain per-bin max-rel -- for every size but
the six of FLOAT32_LIMIT below, where the CPU path itself is ~1e-6 or more from float64 truth and which have a test
of their own that says what is asserted instead; and, since the third stream (round 5), with test_gpu_parity.holds_the_bar's
one per-stream exception: where the CPU path is 9e-7 or more from the truth on THAT stream, the GPU is held to 5e-7 from
the truth instead (63000 bins on held_out_c: CPU path 1.11e-6 from the truth, GPU 2.3e-7).

Round 4's outcome: ten split-form sizes failed and left the table (52000, 64000, 72000, 75000, 76000, 77000, 90000,
98304, 100000, 105000); the first seven passed here on large Bluestein, the last three are in FLOAT32_LIMIT.

Round 5: tools/analysis/parity_passes.cpp found where the split / paired forms lose their accuracy -- the LAST pass, in
which a line's energy has collected in one butterfly and every float32 rounding inside it lands on the weak bins beside
the line -- and the forms of 40000 bins and more run that pass in double now (mixed_core.h, split_is_wide).  The ten
sizes are back in the table ON THE PLANS THEY HAD; a THIRD held-out stream (held_out_c, a new constant below) was
added after the change.  Same rule: what fails here leaves.

The errors are written to the file $RPF_PARITY_RECORD names (tools/gpu_final_check.sh sets it ->
profiles/rNN_fullsize_errors.json, round 5: r05), else to pytest's tmp_path: running the tests has no side effect on the tree."""
import json
import os

import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from helpers import max_err_over_mean, max_rel, oracle_accumulate, truth_f64
from test_gpu_parity import PARITY, THIN_MARGIN_SIZES, holds_the_bar, run_device, torch_dev  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

# The held-out key.  Do not use it, or the two formulas below, anywhere under tools/.
HELD_OUT_KEY = 0x52304F34_48454C44


HELD_OUT_KEY_C = 0x52303543_54484952      # round 5's third stream


def held_out_seeds(N):
    a = (HELD_OUT_KEY ^ (N * 0x9E3779B97F4A7C15)) & 0x7FFFFFFF
    b = ((HELD_OUT_KEY >> 17) + 7919 * N * N + 104729) & 0x7FFFFFFF
    c = ((HELD_OUT_KEY_C * (2 * N + 1)) ^ (HELD_OUT_KEY_C >> 29) ^ (N << 7)) & 0x7FFFFFFF
    return [("held_out_a", a), ("held_out_b", b), ("held_out_c", c)]


# Sizes at which float32 itself gives out on these streams: the CPU path -- the reference's own arithmetic -- sits 0.9e-6
# ... 2.2e-6 from float64 truth in its worst bin (deterministic lines 1e4 above the weakest bins, 17 - 19 butterfly stages),
# so two correct float32 transforms differ by more than 1e-6 per bin whoever computes them.  Measured on the held-out
# streams (profiles/r04_heldout_alternatives.txt): no kernel this library has -- split form, large Bluestein, four-step
# fused or two-kernel -- holds the plain bar AGAINST the CPU path there.  They get the limit test below instead.
# 131072 joined them in round 5: on its four held-out cases the CPU path is 1.05 - 1.49e-6 from the truth (the GPU
# 0.97 - 1.60e-6) -- it held the plain bar against the CPU path in round 4 with no margin (9.0e-7), by the luck of two
# streams.  From 98304 bins up, on tone-rich input, "<= 1e-6 against the CPU path" is not a property of float32.
FLOAT32_LIMIT = {98304: 1e-6, 100000: 1e-6, 105000: 1e-6, 131072: 2.5e-6, 262144: 3e-6, 524288: 3e-6}     # N -> bound on |gpu - truth| / truth

# every size a picker chose a plan for, in any round (the seven sizes that spent round 4 on large Bluestein included)
PICKED_SIZES = sorted(set(n for n in THIN_MARGIN_SIZES if n not in FLOAT32_LIMIT) | {52000, 64000, 72000, 75000, 76000, 77000, 90000})


def record(tmp_path, name, N, out):
    path = os.environ.get("RPF_PARITY_RECORD") or str(tmp_path / "fullsize_errors.json")
    try:
        data = json.load(open(path))
    except Exception:
        data = {}
    data.setdefault(name, {}).setdefault(str(N), {}).update(out)
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def errors_on(N, seed, torch_dev, R=64):
    from oracle import pocketfft_probe          # recorded, not asserted: a float32 FFT nobody here wrote, as a second comparator
    stream = rpf.synth.noise_tones_iq(seed, N * R)
    out = {}
    for windowed in (False, True):
        w = rpf.synth.hann_window(N) if windowed else None
        with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=R), w) as ds:
            got, n = run_device(ds, stream, R, torch_dev)
        assert n == R
        o32, _ = oracle_accumulate(N, stream, R, w, 32)
        truth = truth_f64(N, stream, R, w)
        e = {"gpu_vs_oracle": max_rel(got, o32), "gpu_vs_truth": max_rel(got, truth), "oracle_vs_truth": max_rel(o32, truth)}
        if pocketfft_probe.available():
            pocket, _ = pocketfft_probe.accumulate(N, stream, R, w)
            e.update({"gpu_vs_pocketfft": max_rel(got, pocket), "pocketfft_vs_truth": max_rel(pocket, truth),
                      "oracle_vs_pocketfft": max_rel(o32, pocket)})
        out["hann" if windowed else "rect"] = e
    return out


@pytest.mark.parametrize("N", PICKED_SIZES)
def test_picked_sizes_hold_the_bar_on_streams_no_picker_has_seen(N, torch_dev, tmp_path):
    """64 frames of the noise + tones stream (deterministic lines 1e4 above the weakest bins: a float32 FFT's
    rounding error is coherent there and does not average down), rectangular and Hann, on three held-out seeds, at
    every split-form / paired-form size and 16384 / 32768: GPU against the CPU path."""
    failures = []
    for name, seed in held_out_seeds(N):
        out = errors_on(N, seed, torch_dev)
        record(tmp_path, name, N, out)
        for k, e in out.items():
            if not holds_the_bar(e):
                failures.append((name, k, e))
    assert not failures, (N, failures)


@pytest.mark.parametrize("N", sorted(FLOAT32_LIMIT))
def test_where_float32_gives_out(N, torch_dev, tmp_path):
    """The sizes of FLOAT32_LIMIT on the held-out tone streams, rectangular and Hann.  Where the CPU path itself is ~1e-6
    or more from the truth in its worst bin, what is asserted of the GPU is its distance from float64 TRUTH, in every
    bin: FLOAT32_LIMIT[N] -- 1e-6 for the three Bluestein sizes (measured 5.0 - 8.4e-7: closer to the truth than the CPU
    path's worst case), 3e-6 at 262144 and 524288 (measured 1.5 - 2.5e-6 where the CPU path has 1.1 - 2.2e-6).  Its
    distance from the CPU path is what it is between two float32 transforms there -- measured 0.6 - 2.0e-6, recorded, and
    held to 2.5e-6 as a sanity bound; on the bins where the CPU path is itself within 5e-7 of the truth (>= 99.9 % of
    them) it is recorded too (measured 4.1e-7 - 1.06e-6)."""
    for name, seed in held_out_seeds(N)[: 1 if N > 262144 else 3]:
        stream = rpf.synth.noise_tones_iq(seed, N * 64)
        for windowed in ((False,) if N > 262144 else (False, True)):
            w = rpf.synth.hann_window(N) if windowed else None
            with rpf.Datastore(rpf.Params(N=N, window=windowed, repeats=64), w) as ds:
                got, n = run_device(ds, stream, 64, torch_dev)
            assert n == 64
            o32, _ = oracle_accumulate(N, stream, 64, w, 32)
            truth = truth_f64(N, stream, 64, w)
            good = np.abs(o32 - truth) < 5e-7 * truth
            e = {"gpu_vs_oracle": max_rel(got, o32), "gpu_vs_truth": max_rel(got, truth), "oracle_vs_truth": max_rel(o32, truth),
                 "gpu_vs_oracle_over_mean": max_err_over_mean(got, o32),
                 "bins_where_cpu_is_within_5e-7_of_truth": float(good.mean()),
                 "gpu_vs_oracle_on_those_bins": float(np.max(np.abs(got[good] - o32[good]) / o32[good]))}
            record(tmp_path, name + "_float32_limit", N, {"hann" if windowed else "rect": e})
            assert e["gpu_vs_truth"] < FLOAT32_LIMIT[N], (N, name, windowed, e)
            assert e["gpu_vs_oracle"] < 2.5e-6 and e["bins_where_cpu_is_within_5e-7_of_truth"] > 0.999, (N, name, windowed, e)
