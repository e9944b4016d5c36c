"""The plan-search tooling (tools/gen_mixed_plans.py, tools/pick_split_plans.py): the candidates it emits are
well-formed plans of the split form, and the picker applies the rule its header states.  CPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_mixed_plans as g          # noqa: E402
import pick_split_plans as pick      # noqa: E402


def test_split_candidates_are_well_formed():
    for n in (14000, 20000, 6250, 64000, 100000, 98304):
        cands = g.split_candidates(n)
        assert cands, n
        for p, m, rad, gs in cands:
            assert p in (2, 3, 4, 5, 6, 8, 10) and p * m == n and m % 2 == 0 and m <= 16384
            prod = 1
            for r, k in zip(rad, gs):
                prod *= r
                assert 2 <= r <= 25 and m % (r * k) == 0 and r * k <= 32
            assert prod == m and len(rad) >= 3
            assert max(m // (r * k) for r, k in zip(rad, gs)) <= 1024           # one frame slot per workgroup
            assert g.lds_bytes(m, rad, gs, 1, 2) <= g.LDS_LIMIT


def test_window_candidates_name_a_mode_the_kernel_has():
    for n in (20000, 50000, 16384):
        for p, m, rad, gs, wm in g.window_candidates(n):
            assert wm in (2, 3) and p * m == n
            if wm == 3:                  # the whole window beside the slab
                assert g.lds_bytes(m, rad, gs, 1, 2) + 4 * n <= g.LDS_LIMIT


def test_picker_rule():
    fast_inaccurate = (300.0, 9e-7, "a")
    soft = (290.0, 5.3e-7, "b")
    ok = (180.0, 4.0e-7, "c")
    ok_fast = (280.0, 4.9e-7, "d")
    assert pick.choose([fast_inaccurate, soft, ok]) == soft              # 1.5 x faster within the soft limit
    assert pick.choose([fast_inaccurate, soft, ok_fast, ok]) == ok_fast  # the fastest within the limit
    assert pick.choose([fast_inaccurate, (200.0, 6.2e-7, "e")]) == (200.0, 6.2e-7, "e")      # fallback
    assert pick.choose([fast_inaccurate]) is None


def test_no_tool_knows_the_held_out_streams():
    """tests/test_gpu_heldout.py holds every picked size to the parity bar on streams derived from constants that live
    in that file only.  No plan picker, scorer or generator under tools/ may import it, name it or restate its formulas.
    Two tools read the seeds of the streams a, b, c from there (tuning_stream_seeds) and choose no plan:
    gpu_heldout_alternatives.py measures the kernels that take a failed size BACK (large Bluestein, the two-kernel pair),
    analysis/parity_passes.py measures which pass of a SHIPPED plan loses the accuracy (round 5) -- which made a, b, c
    tuning streams.  Round 6's streams d and e (HELD_OUT_KEY_R6) are named by NO file under tools/, readers included.
    A script may RUN the test file under pytest."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "tests", "test_gpu_heldout.py")).read()
    keys = dict(re.findall(r"(HELD_OUT_KEY\w*) = (0x[0-9A-Fa-f_]+)", text))
    assert sorted(keys) == ["HELD_OUT_KEY", "HELD_OUT_KEY_C", "HELD_OUT_KEY_R6"]
    readers = {"gpu_heldout_alternatives.py", os.path.join("analysis", "parity_passes.py")}
    for path in glob.glob(os.path.join(root, "tools", "**", "*"), recursive=True):
        rel = os.path.relpath(path, os.path.join(root, "tools"))
        if not os.path.isfile(path) or path.endswith((".so", ".pyc")):
            continue
        try:
            body = open(path, errors="ignore").read()
        except OSError:
            continue
        low = body.lower()
        for key in keys.values():
            assert key.lower() not in low and key.replace("_", "").lower() not in low, path
        # round 6's streams: nobody under tools/, not even the two readers
        for word in ("HELD_OUT_KEY_R6", "held_out_d", "held_out_e", "held_out_seed(", "_mix("):
            assert word not in body, (path, word)
        if rel in readers:
            assert "pick" not in low.replace("picks nothing", "") and "mixed_plans" not in body, rel
            continue
        assert "tuning_stream_seeds" not in body and "HELD_OUT_KEY" not in body, path
        for line in body.split("\n"):
            assert "test_gpu_heldout" not in line or "pytest" in line, (path, line)


def test_lds_bank_model_of_the_fused_kernel():
    """tools/lds_bank_model.py: the numbers profiles/r04_c4_fused.txt 6. quotes, and the model's own index maps --
    Geom.bin_of is a bijection with the additive split the fused kernel's twiddle factorisation needs, the dropped
    staging swizzles are permutations that keep a pair of bins adjacent."""
    import lds_bank_model as m
    for N in (512, 256, 128):
        g = m.Geom(N)
        bins = sorted(g.bin_of(t, a) for t in range(g.T) for a in range(g.P))
        assert bins == list(range(N))
        assert all(g.bin_of(t, a) == g.bin_of(t, 0) + g.bin_of(0, a) for t in range(g.T) for a in range(g.P))
        for J in range(1, g.NPASS + 1):
            assert sorted(g.elem(J, t, a) for t in range(g.T) for a in range(g.P)) == list(range(N))
        assert m.exchange(N) == (128, 96)                               # first-pass stores take twice their cycles
        assert m.staging(N) == (64, 32, 64, 32)                         # both directions twice
        sw = m.SWIZZLE[N]
        assert sorted(sw(k) for k in range(N)) == list(range(N))
        assert m.staging(N, sw, N, wide=True) == (32, 32, 16, 16)       # conflict-free -- and measured worth nothing
    assert [m.raw_rows(b, False)[0] for b in (32, 64, 128)] == [16, 32, 64]
    assert [m.raw_rows(b, True)[0] for b in (32, 64, 128)] == [8, 8, 8]


def test_why_the_four_step_kernels_multiply_by_exact_twiddles():
    """tools/analysis/fourstep_passes.py, the float32 emulation rpf_fourstep.hip's round-6 passes were designed on, at
    131072 bins on the pickers' stream (64 frames, CPU, ~10 s): with the row transform's last pass in double the worst
    bin is still ~1e-6 from float64 truth; making nothing but the pass-before-the-last's twiddles exact (butterflies
    float32: the shipped form) halves that; the GPU's own numbers are profiles/r06_fourstep_wide.txt."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "analysis"))
    import fourstep_passes as fp
    case = fp.Case(131072, R=64)
    err = {name: case.error(**kw)[0] for name, kw in fp.FORMS.items() if name != "float32"}
    assert err["last_pass_double"] > 8e-7, err
    assert err["exact_twiddles"] < 0.6 * err["last_pass_double"], err
    assert err["last_two_passes_double"] <= err["exact_twiddles"] * 1.05, err
