#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- float64 ground truth for the hot path.

The reference ships no golden vectors and cannot be built in this
image (it needs fftw3/TCLAP/librtlsdr), so these fixtures are computed from the
mathematical definition of the path (/root/reference/src/datastore.cxx:66-89):
exact unpack arithmetic in float32 (integers, and one float32 rounding for the
window product), then a complex128 numpy FFT and a float64 sum of |X|^2.
Inputs come from the integer-only generators in rtl-power-fftw_amd/synth.py and
are regenerated from their seed by the tests, so only the expected spectra are
stored.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import rtl_power_fftw_amd as rpf  # noqa: E402

CASES = [
    # name, N, repeats, generator, seed, window
    ("c1_n512_r100_uniform", 512, 100, "uniform", 1, False),       # BASELINE.json configs[0]
    ("n512_r100_hann", 512, 100, "noise_tones", 11, True),
    ("n4096_r64_noise", 4096, 64, "noise_tones", 2, False),        # head of config C2's stream
    ("n4096_r64_hann", 4096, 64, "noise_tones", 2, True),          # head of config C3's stream
    ("n64_r33_uniform", 64, 33, "uniform", 3, False),
    ("n1024_r17_noise", 1024, 17, "noise_tones", 4, True),
    ("n8192_r9_noise", 8192, 9, "noise_tones", 5, False),
    ("n500_r20_uniform", 500, 20, "uniform", 6, False),            # man page's non-power-of-two example size
    # config C4's size; noise only: at this N the weak tones of noise_tones would
    # tower 3e4 x over the noise floor and no float32 FFT resolves the floor bins
    # of such a frame to 1e-6
    ("n262144_r2_uniform", 262144, 2, "uniform", 7, False),
    ("n16384_r8_uniform", 16384, 8, "uniform", 8, True),           # smallest four-step size, windowed
    ("n5000_r12_uniform", 5000, 12, "uniform", 9, False),          # large Bluestein path, M = 16384
    ("n100000_r3_uniform", 100000, 3, "uniform", 10, False),       # large Bluestein path, M = 262144
    # BASELINE.json configs[3] on its OWN stream ("as C2, seed 4", tones included), all
    # 1000 frames: every 64th bin, +-512 bins around the two tone lines, and the total
    ("c4_n262144_r1000_seed4", 262144, 1000, "noise_tones", 4, False),
] + [
    # BASELINE.json configs[4]: 8 hops x 5000 frames of N = 4096, hop h has seed 50 + h;
    # every hop has its own pwr (acquisition.cxx:252-254), so one full spectrum per hop
    ("c5_hop%d_n4096_r5000" % h, 4096, 5000, "noise_tones", 50 + h, False) for h in range(8)
]


def stream_for(gen, seed, nsamples):
    if gen == "uniform":
        return rpf.synth.uniform_iq(seed, nsamples)
    return rpf.synth.noise_tones_iq(seed, nsamples)


def truth(N, stream, repeats, window, chunk_bytes=1 << 28):
    """datastore.cxx:66-89 in float64, a bounded number of frames at a time."""
    sign = (1 - 2 * (np.arange(N) % 2)).astype(np.float32)
    pwr = np.zeros(N)
    step = max(1, chunk_bytes // (16 * N))
    for f0 in range(0, repeats, step):
        f1 = min(repeats, f0 + step)
        x = stream[2 * N * f0: 2 * N * f1].astype(np.float32).reshape(f1 - f0, N, 2) - np.float32(127.0)
        x = x * sign[None, :, None]
        if window is not None:
            x = x * window[None, :, None]          # float32 product, one rounding
        z = x[..., 0].astype(np.float64) + 1j * x[..., 1].astype(np.float64)
        spec = np.fft.fft(z, axis=1)
        pwr += (spec.real ** 2 + spec.imag ** 2).sum(axis=0)
    return pwr


def tone_bins(N):
    """Bins of noise_tones_iq's two lines after the (-1)^n centring: the period-8
    tone (N/8) and the period-16 tone stepped 3 samples at a time (3N/16)."""
    return [(N // 8 + N // 2) % N, (3 * N // 16 + N // 2) % N]


def main():
    only = set(sys.argv[1:])          # optional: regenerate just these fixtures
    for name, N, R, gen, seed, win in CASES:
        if only and name not in only:
            continue
        stream = stream_for(gen, seed, N * R)
        window = rpf.synth.hann_window(N) if win else None
        pwr = truth(N, stream, R, window)
        out = dict(N=N, repeats=R, generator=gen, seed=seed, pwr=pwr,
                   stream_crc=np.uint64(int(np.bitwise_xor.reduce(stream.view(np.uint8).astype(np.uint64) * np.arange(1, stream.size + 1, dtype=np.uint64)))))
        if window is not None:
            out["window"] = window
        if N > 16384:
            # keep the fixture small: every 64th bin plus the total
            out["pwr"] = pwr[::64].copy()
            out["stride"] = 64
            out["total"] = pwr.sum()
            if gen == "noise_tones":
                # ... and the neighbourhood of the two lines, where a float32 FFT is under most stress
                near = np.unique(np.concatenate([np.arange(b - 512, b + 513) % N for b in tone_bins(N)]))
                out["near_bins"] = near.astype(np.int64)
                out["near_pwr"] = pwr[near].copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, N, R, "sum %.6e" % pwr.sum())


if __name__ == "__main__":
    main()
