"""Which pass of the four-step transform carries the error that does not average down?  (CPU only, numpy.)

A float32 emulation of X[k1 + N1 k2] = sum_n2 W_N2^{n2 k2} ( W_N^{n2 k1} sum_n1 x[N2 n1 + n2] W_N1^{n1 k1} ) -- decimation in
frequency, radix-8 passes like rpf_fourstep.hip's, every pass's outputs rounded to float32 -- on 64 frames of the noise +
tones stream, with any pass run in double instead, and with the twiddle TABLES either rounded to float32 (what the kernels
read) or exact.  It is not bit-identical to the kernels (the butterflies here are direct sums); it answers WHICH rounding
matters.  Round 6's finding (profiles/r06_fourstep_wide.txt): at 131072 bins the worst bin is a weak deterministic line
(bin 2N/16) that shares its last two row butterflies with a strong line; arithmetic in double alone leaves 1.3e-6 there,
because the float32 REPRESENTATION error of the pass-2 twiddles (1.7e-8 on 1/sqrt(2)) leaks the strong line into the weak
one identically in every frame; the last two row passes in double WITH exact twiddles leave 2.3e-7, and exact twiddles
alone -- the pass's butterflies still float32, each product x (hi + lo) rounded once -- 2.7 - 5.3e-7 (the shipped form).

Usage: python tools/analysis/fourstep_passes.py N [seed]     (N = 65536, 131072, 262144; default seed: the pickers' 300 + N % 89)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import rtl_power_fftw_amd as rpf
from helpers import truth_f64


def dft_mat(R, dtype):
    k = np.arange(R)
    return np.exp(-2j * np.pi * np.outer(k, k) / R).astype(dtype)


def butterfly8_float32(x, exact_const):
    """x[..., 8] complex64 -> the 8-point DFT over the last axis in float32 arithmetic, stage by stage like Dft<8> of
    dft_small.h (radix-2 stage, the W8 products, two radix-4 butterflies).  The two products by (1 -+ i)/sqrt(2) use the
    float32 constant, or (exact_const) the exact one with ONE rounding of each product -- what a (hi + lo) constant gives."""
    f32 = np.float32
    v = [x[..., i].astype(np.complex64) for i in range(8)]
    a = [v[i] + v[i + 4] for i in range(4)]
    b = [v[i] - v[i + 4] for i in range(4)]
    c = 1 / np.sqrt(2.0) if exact_const else f32(1 / np.sqrt(2.0))

    def scaled(re, im):
        if exact_const:
            return ((re.astype(np.float64) * c).astype(f32) + 1j * (im.astype(np.float64) * c).astype(f32)).astype(np.complex64)
        return ((re * c).astype(f32) + 1j * (im * c).astype(f32)).astype(np.complex64)

    def minus_i(z):
        return (z.imag - 1j * z.real).astype(np.complex64)
    b = [b[0], scaled((b[1].real + b[1].imag).astype(f32), (b[1].imag - b[1].real).astype(f32)), minus_i(b[2]),
         scaled((b[3].imag - b[3].real).astype(f32), (-(b[3].real + b[3].imag)).astype(f32))]

    def radix4(u):
        t0, t1, t2, t3 = u[0] + u[2], u[0] - u[2], u[1] + u[3], minus_i(u[1] - u[3])
        return [(t0 + t2).astype(np.complex64), (t1 + t3).astype(np.complex64), (t0 - t2).astype(np.complex64),
                (t1 - t3).astype(np.complex64)]
    even, odd = radix4(a), radix4(b)
    out = np.empty(v[0].shape + (8,), dtype=np.complex64)
    for k in range(4):
        out[..., 2 * k], out[..., 2 * k + 1] = even[k], odd[k]
    return out


def fft_dif(x, radices, wide=(), exact_tw=(), comp_tw=(), last8=None):
    """x: [B, L] complex64 -> natural-order spectrum [B, L] complex64.  Pass i (radix radices[i]) runs in complex128 when
    i is in `wide` (inputs converted, outputs rounded back to complex64); its output twiddles are the float32-rounded
    table's values unless i is in `exact_tw` (a wide pass with double twiddles) or in `comp_tw` (a float32 pass whose
    products are x (hi + lo): evaluated here as the float32 butterfly output times the exact twiddle, rounded once).
    last8 = "float32" / "exact_const": a last pass of radix 8 is butterfly8_float32 (explicit float32 steps) instead."""
    B, L = x.shape
    cur = x.reshape(B, 1, L)
    for i, R in enumerate(radices):
        G, Ls = cur.shape[1], cur.shape[2]
        M = Ls // R
        dt = np.complex128 if i in wide else np.complex64
        if last8 is not None and i == len(radices) - 1 and R == 8:
            y = butterfly8_float32(cur.reshape(B, G, R, M).transpose(0, 1, 3, 2), last8 == "exact_const").transpose(0, 1, 3, 2)
        else:
            y = np.einsum("ra,bgam->bgrm", dft_mat(R, dt), cur.reshape(B, G, R, M).astype(dt))
        if M > 1:
            tw = np.exp(-2j * np.pi * np.outer(np.arange(R), np.arange(M)) / Ls)
            if i in comp_tw:      # float32 butterfly, product with the exact twiddle rounded once (a hi + lo float pair)
                y = (y.astype(np.complex64).astype(np.complex128) * tw[None, None]).astype(np.complex64)
            else:
                tw = tw.astype(dt) if i in exact_tw else tw.astype(np.complex64).astype(dt)
                y = y * tw[None, None]
        cur = y.astype(np.complex64).reshape(B, G * R, M)
    out = cur.reshape(B, L)
    # group index = digits (r1, r2, ...) most significant first; bin = r1 + R1 r2 + R1 R2 r3 ...
    rem = np.arange(L)
    k = np.zeros(L, dtype=np.int64)
    weight = 1
    for j, R in enumerate(radices):
        span = int(np.prod(radices[j + 1:])) if j + 1 < len(radices) else 1
        k += (rem // span) * weight
        rem = rem % span
        weight *= R
    res = np.empty_like(out)
    res[:, k] = out
    return res


def fourstep(x, N1, N2, rad1, rad2, wide_col=(), wide_row=(), exact_row=(), exact_col=(), step_exact=False, comp_row=(),
             last8=None):
    B = x.shape[0]
    cols = np.ascontiguousarray(x.reshape(B, N1, N2).transpose(0, 2, 1)).reshape(B * N2, N1)
    A = fft_dif(cols, rad1, wide_col, exact_col).reshape(B, N2, N1)
    tw = np.exp(-2j * np.pi * np.outer(np.arange(N2), np.arange(N1)) / (N1 * N2))
    A = (A.astype(np.complex128) * tw[None]).astype(np.complex64) if step_exact else A * tw.astype(np.complex64)[None]
    rows = np.ascontiguousarray(A.transpose(0, 2, 1)).reshape(B * N1, N2)
    X = fft_dif(rows, rad2, wide_row, exact_row, comp_row, last8).reshape(B, N1, N2)
    return np.ascontiguousarray(X.transpose(0, 2, 1)).reshape(B, N1 * N2)


SPLITS = {65536: (256, 256, (8, 8, 4), (8, 8, 4)), 131072: (512, 256, (8, 8, 8), (8, 8, 4)),
          262144: (512, 512, (8, 8, 8), (8, 8, 8))}


class Case:
    """R frames of the noise + tones stream of `seed` at N bins, rectangular: the unpacked samples and the float64 truth."""

    def __init__(self, N, seed=None, R=64):
        self.N, self.R = N, R
        self.seed = 300 + N % 89 if seed is None else seed      # default: the stream the plan pickers scored on
        stream = rpf.synth.noise_tones_iq(self.seed, N * R)
        self.truth = truth_f64(N, stream, R)
        x = stream.astype(np.float32).reshape(R, N, 2) - np.float32(127)
        x = x * (1 - 2 * (np.arange(N) % 2)).astype(np.float32)[None, :, None]
        self.x = (x[..., 0] + 1j * x[..., 1]).astype(np.complex64)

    def error(self, **kw):
        """(max over bins of |pwr - truth| / truth, that bin, the 99.9 % quantile) of the emulated transform `kw`"""
        N1, N2, r1, r2 = SPLITS[self.N]
        acc = np.zeros(self.N)
        for f0 in range(0, self.R, 8):
            X = fourstep(self.x[f0:f0 + 8], N1, N2, r1, r2, **kw).astype(np.complex128)
            acc += (X.real ** 2 + X.imag ** 2).sum(0)
        rel = np.abs(acc - self.truth) / self.truth
        return float(rel.max()), int(np.argmax(rel)), float(np.quantile(rel, 0.999))


# the forms rpf_fourstep.hip can be built in (RPF_FOURSTEP_WIDE / RPF_FOURSTEP_WIDE2), as arguments of fourstep()
FORMS = {
    "float32": dict(),
    "last_pass_double": dict(wide_row=(2,)),
    "exact_twiddles": dict(wide_row=(2,), comp_row=(1,)),                 # shipped from 131072 bins up
    "last_two_passes_double": dict(wide_row=(1, 2), exact_row=(1,)),
}


def main():
    N = int(sys.argv[1])
    case = Case(N, int(sys.argv[2]) if len(sys.argv) > 2 else None)
    N1 = SPLITS[N][0]
    print("N = %d = %d x %d, seed %d, %d frames, rectangular; max over bins of |pwr - truth| / truth" % (
        N, N1, SPLITS[N][1], case.seed, case.R))

    def run(name, **kw):
        t0 = time.time()
        worst, k, q = case.error(**kw)
        print("%-62s %.2e at bin %6d (k1 = %3d, k2 = %3d); 99.9 %% of bins below %.2e  [%.0f s]" % (
            name, worst, k, k % N1, k // N1, q, time.time() - t0), flush=True)

    run("every pass float32, float32 twiddle tables (rounds 1 - 5)")
    run("last row pass in double", wide_row=(2,))
    run("last two row passes in double, float32 tables", wide_row=(1, 2))
    run("every pass of both transforms in double, float32 tables", wide_row=(0, 1, 2), wide_col=(0, 1, 2), step_exact=True)
    run("every pass float32, EXACT tables", exact_row=(0, 1), exact_col=(0, 1), step_exact=True)
    run("last row pass in double, pass-2 row twiddles exact", wide_row=(2,), exact_row=(1,))
    run("last row pass in double, pass-2 float32 butterfly x (hi + lo) twiddle (shipped from 131072)", wide_row=(2,), comp_row=(1,))
    run("last two row passes in double, pass-2 row twiddles exact (RPF_FOURSTEP_WIDE2=2)", wide_row=(1, 2), exact_row=(1,))
    if SPLITS[N][3][-1] == 8:
        # what the double last pass buys when it is a radix-8 pass: mostly an exact 1/sqrt(2)
        run("float32 last pass, explicit steps (= every pass float32)", last8="float32")
        run("float32 last pass with an exact 1/sqrt(2), float32 tables", last8="exact_const")
        run("float32 last pass with an exact 1/sqrt(2), pass-2 (hi + lo) twiddles", last8="exact_const", comp_row=(1,))
    run("every pass in double, EXACT tables (float32 storage only)", wide_row=(0, 1, 2), wide_col=(0, 1, 2), exact_row=(0, 1),
        exact_col=(0, 1), step_exact=True)


if __name__ == "__main__":
    main()
