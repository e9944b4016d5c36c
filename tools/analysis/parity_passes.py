#!/usr/bin/env python3
"""Which stage of the split / paired mixed-radix form costs the accuracy? (CPU only)

For a shipped split-form plan: the spectrum of 64 frames of a held-out tone stream with every stage as shipped, with
ONE stage replaced by its ideal twin (exact arithmetic, exact twiddles, one rounding per element), with one stage's
twiddle products compensated, and with everything ideal -- each against float64 truth and against the CPU oracle.
The held-out streams are named by tests/test_gpu_heldout.py (imported from there: the key is not copied).

usage: parity_passes.py N[:w] ...      (w: the windowed twin under a Hann window)
       parity_passes.py --radix        the small butterflies' own rounding error, radix by radix: on random inputs, and
                                       beside a line 100 x the other outputs (what the last pass sees)
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rtl_power_fftw_amd as rpf  # noqa: E402
from helpers import max_rel, oracle_accumulate, truth_f64  # noqa: E402
from test_gpu_heldout import tuning_stream_seeds  # noqa: E402


def load():
    so = os.path.join(HERE, "libparity_passes.so")
    src = os.path.join(HERE, "parity_passes.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                        "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.rpf_analysis_split.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    return lib


def run(lib, N, twin, window, stream, R, mode):
    pwr = np.zeros(N)
    m = (ctypes.c_int * 8)(*(list(mode) + [0] * (8 - len(mode))))
    rc = lib.rpf_analysis_split(N, twin, window.ctypes.data if window is not None else None, stream.ctypes.data, R, m,
                                pwr.ctypes.data)
    assert rc > 0, (N, twin, rc)
    return pwr, rc


def radix_tables(lib):
    out = (ctypes.c_double * 2)()
    lib.rpf_analysis_radix_error.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    lib.rpf_analysis_line_exposure.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    eps = 2.0 ** -24
    print("SmallDft<R> (dft_small.h), float32; eps = 2^-24")
    print(" R   random inputs: rms error / rms output   | one output 100 x the others: rms error on the WEAK outputs, in eps x line")
    print("                                             |   the butterfly's arithmetic    the rounding of its float inputs")
    for R in range(2, 26):
        lib.rpf_analysis_radix_error(R, 20000, out)
        a = out[0] / eps
        lib.rpf_analysis_line_exposure(R, 20000, 100.0, out)
        print("%2d   %.2f eps                                   |   %.3f                          %.3f" % (R, a, out[0], out[1]))


def main():
    lib = load()
    if sys.argv[1:] == ["--radix"]:
        radix_tables(lib)
        return
    R = int(os.environ.get("FRAMES", "64"))
    for arg in sys.argv[1:]:
        N = int(arg.split(":")[0])
        windowed = arg.endswith(":w")
        window = rpf.synth.hann_window(N) if windowed else None
        only_stream = os.environ.get("STREAM")        # a, b or c: that held-out stream only
        seeds = [sd for sd in tuning_stream_seeds(N) if not only_stream or sd[0].endswith("_" + only_stream)]
        for name, seed in seeds[: int(os.environ.get("STREAMS", "3"))]:
            stream = rpf.synth.noise_tones_iq(seed, N * R)
            truth = truth_f64(N, stream, R, window)
            o32, _ = oracle_accumulate(N, stream, R, window, 32)
            _, F = run(lib, N, int(windowed), window, stream, 1, [0] * 8)
            nst = F + 1
            rows = [("as shipped", [0] * nst)]
            for s in range(nst):
                rows.append(("stage %d ideal" % s, [2 if i == s else 0 for i in range(nst)]))
            for s in range(nst - 1):
                rows.append(("stage %d twiddles compensated" % s, [3 if i == s else 0 for i in range(nst)]))
            rows.append(("all twiddles compensated", [3] * (nst - 1) + [0]))
            rows.append(("last stage: exact arithmetic on float DFT constants", [0] * (nst - 1) + [4]))
            rows.append(("last stage wide (double butterfly, as the kernels can run it)", [0] * (nst - 1) + [5]))
            for st in range(nst - 1):
                rows.append(("last stage wide + stage %d ideal" % st, [2 if i == st else 0 for i in range(nst - 1)] + [5]))
            rows.append(("last stage wide + the one before it in exact arithmetic on float twiddles", [0] * (nst - 2) + [1, 5]))
            rows.append(("all exact arithmetic, float twiddles", [1] * nst))
            rows.append(("all ideal", [2] * nst))
            print("N = %d %s  %s  (M-point plan: %d passes; oracle vs truth %.2e)" % (
                N, "hann" if windowed else "rect", name, F, max_rel(o32, truth)))
            only = os.environ.get("ONLY")
            if only:
                rows = [r for r in rows if any(o in r[0] for o in only.split(","))]
            for label, mode in rows:
                got, _ = run(lib, N, int(windowed), window, stream, R, mode)
                print("   %-40s vs truth %.2e   vs oracle %.2e" % (label, max_rel(got, truth), max_rel(got, o32)), flush=True)


if __name__ == "__main__":
    main()
