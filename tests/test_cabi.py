"""The C-ABI library loads and exports exactly what include/rpf_engine.h declares;
argument validation works without a GPU; without a device the engine fails loudly
(no CPU fallback).  (-m "not gpu")"""
import ctypes
import os
import re

import numpy as np
import pytest

import rtl_power_fftw_amd as rpf
from rtl_power_fftw_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(path=os.path.join(ROOT, "include", "rpf_engine.h")):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rpf_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.symbol_names())


def test_test_hooks_are_not_part_of_the_boundary():
    """The fault-injection entry lives in csrc/rpf_engine_testing.h: exported for tests/, absent from the ABI header,
    and no host code (the C++ mirror of the reference's classes, the CLI) includes that header or calls it."""
    hooks = _declared_symbols(os.path.join(ROOT, "rtl-power-fftw_amd", "csrc", "rpf_engine_testing.h"))
    assert hooks == sorted(_lib.test_hook_names()) == ["rpf_debug_fused_fault"]
    assert not any("debug" in name for name in _declared_symbols())
    lib = ctypes.CDLL(_lib.lib_path())
    for name in hooks:
        assert hasattr(lib, name), name
    host = os.path.join(ROOT, "rtl-power-fftw_amd", "host")
    for f in os.listdir(host):
        if f.endswith((".cpp", ".h")):
            body = open(os.path.join(host, f)).read()
            assert "rpf_engine_testing" not in body and "rpf_debug_" not in body, f


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.lib_path())
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert rpf.load().rpf_abi_version() == 2


def test_supported_sizes():
    lib = rpf.load()
    for n in (2, 6, 32, 64, 128, 256, 500, 512, 1000, 1024, 2046, 2048, 3000, 4094, 4096, 8192, 16384,
              32768, 65536, 131072, 262144, 4098, 5000, 16386, 100000, 131070,
              131074, 200000, 524288, 999998, 1048576, 1 << 22, 1 << 24, 3000000, 1 << 23, 1 << 26):      # catch-all Stockham path
        assert lib.rpf_supported_n(n) == 1
    for n in (0, 1, 513, (1 << 23) + 2, 10000000, 1 << 27):
        assert lib.rpf_supported_n(n) == 0


def test_invalid_arguments_map_to_reference_exit_codes():
    # exit codes of /root/reference/src/exceptions.h:25-34
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=10000000))                 # no kernel for this size
    assert e.value.retval == rpf.ReturnValue.InvalidArgument
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=511))                      # odd (params.cxx:150-155 bumps it upstream)
    assert e.value.retval == rpf.ReturnValue.InvalidArgument
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=512, buffers=0))
    assert e.value.retval == rpf.ReturnValue.InvalidArgument
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=512, window=True), window_values=np.ones(5, dtype=np.float32))
    assert e.value.retval == rpf.ReturnValue.InvalidInput
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=4096), flags=(200 << 8))    # unknown tuning variant
    assert e.value.retval == rpf.ReturnValue.InvalidArgument


def test_shipped_library_has_no_tuning_or_ablation_variants():
    """One kernel per N in the product: the experimental variants (ablations included)
    exist only in the -DRPF_TUNING build, and RPF_FLAG_VARIANT(k != 0) is an error."""
    if os.environ.get("RPF_ENGINE_LIB"):
        pytest.skip("an alternative build is loaded")
    for n in (512, 1024, 2048, 4096, 8192):
        for vid in range(1, 64):
            with pytest.raises(rpf.RPFError) as e:
                rpf.Datastore(rpf.Params(N=n), flags=(vid << 8))
            assert e.value.retval == rpf.ReturnValue.InvalidArgument


def test_no_device_means_hardware_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rpf.RPFError) as e:
        rpf.Datastore(rpf.Params(N=4096))
    assert e.value.retval == rpf.ReturnValue.HardwareError
    assert "no CPU path" in str(e.value)


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["readelf", "-d", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "rpf_oracle" not in out and "rpf_emul" not in out
    syms = subprocess.run(["nm", "-D", _lib.lib_path()], capture_output=True, text=True).stdout
    assert "rpf_oracle" not in syms


def test_shipped_mixed_radix_plans_are_well_formed():
    """mixed_plans.inc / mixed_plans_split.inc (what tools/pick_mixed_plans.py and a human wrote): radices multiply to
    the length, butterflies per thread divide it, the workgroup and its LDS fit the hardware, no size twice, and the
    engine reports every planned size as supported."""
    import re
    lib = rpf.load()
    seen = set()
    for name in ("mixed_plans.inc", "mixed_plans_split.inc"):
        text = open(os.path.join(ROOT, "rtl-power-fftw_amd", "csrc", name)).read().split("#ifdef RPF_TUNING")[0]
        macros = dict(re.findall(r"#define (RPF_M\w+) (MixPlan<.*>)", text))
        for split, plan, variant in re.findall(r"(?:plan|split)_entry<(?:(\d+), )?(RPF_M\w+|MixPlan<[^()]*?>)(?:, \d, (?:true|false))?>\((\d+)\)", text):
            plan = macros.get(plan, plan)
            m = re.match(r"MixPlan<(\d+), (\d+), (\d+), (.*)>$", plan)
            assert m, plan
            length, fpw, tw = int(m.group(1)), int(m.group(2)), int(m.group(3))
            passes = [(int(r), int(g or 1)) for r, g in re.findall(r"P<(\d+)(?:, (\d+))?>", m.group(4))]
            prod = 1
            for r, g in passes:
                prod *= r
                assert 2 <= r <= 25 and length % (r * g) == 0, plan
            assert prod == length and len(passes) >= 2 and int(variant) == 0, plan
            tpf = max(length // (r * g) for r, g in passes)
            assert fpw * tpf <= 1024 and tw in (0, 1, 2), plan
            rlast = passes[-1][0]
            cpx = length + length // rlast if rlast % 2 == 0 else length
            table = 0
            if tw == 1:
                table = sum(g * (r - 1) * (length // (r * g)) for r, g in passes[:-1])
            elif tw == 2:
                s = length
                for i, (r, g) in enumerate(passes[:-1]):
                    s //= r
                    if i >= 1:
                        table += (r - 1) * s
            assert (fpw * cpx + table) * 8 <= 160 * 1024, plan
            n = length * int(split or 1)
            if split:
                assert fpw == 1 and tw != 1 and length % 2 == 0 and int(split) in (2, 3, 4, 5, 6, 8, 10), plan
            assert n not in seen, n
            seen.add(n)
            assert lib.rpf_supported_n(n) == 1, n
    assert len(seen) >= 199 and {500, 1000, 7000, 10000, 14000, 16384, 20000, 32768, 50000, 60000, 80000, 96000} <= seen
    # (round 4 took ten split-form sizes out; round 5 put them back on the plans they had, with the last pass in double:
    #  tests/test_gpu_heldout.py decides whether they stay)
    assert {52000, 64000, 72000, 75000, 76000, 77000, 90000, 98304, 100000, 105000} <= seen and len(seen) >= 209
    # the per-size overrides (find_form takes the first match): sizes of the tables, each (size, run) once, split form
    text = open(os.path.join(ROOT, "rtl-power-fftw_amd", "csrc", "mixed_plans_override.inc")).read()
    keys = re.findall(r"^\s+\{(\d+), (true|false), split_form<(\d+), ", text, flags=re.M)
    assert len(keys) > 50 and len(set((n, w) for n, w, _ in keys)) == len(keys)
    assert all(int(n) in seen and int(n) % int(p) == 0 and int(p) in (2, 3, 4, 5, 6, 8, 10) for n, _, p in keys)
