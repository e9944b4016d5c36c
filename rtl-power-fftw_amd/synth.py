"""Synthetic RTL-SDR byte streams (interleaved unsigned 8-bit I/Q, offset
binary, as ``rtlsdr_read_sync`` delivers them, /root/reference/src/device.cxx:92-97).

Everything is integer arithmetic on a counter-based splitmix64 generator, so a
(seed, shape) pair yields bit-identical bytes on every machine -- the committed
golden vectors under tests/golden/ depend on that.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, offset=0):
    """n 64-bit words of the splitmix64 stream of `seed`, starting at word `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform_iq(seed, nsamples):
    """Uniformly random bytes: 2*nsamples uint8 (config C1 of SURVEY.md 8d)."""
    nwords = (2 * nsamples + 7) // 8
    return splitmix64(seed, nwords).view(np.uint8)[: 2 * nsamples].copy()


# integer tone tables: round(A*cos(2 pi n / period)), round(A*sin(2 pi n / period))
_TONE8_C = np.array([10, 7, 0, -7, -10, -7, 0, 7], dtype=np.int32)
_TONE8_S = np.array([0, 7, 10, 7, 0, -7, -10, -7], dtype=np.int32)
_TONE16_C = np.array([6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2, 0, 2, 4, 6], dtype=np.int32)
_TONE16_S = np.array([0, 2, 4, 6, 6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2], dtype=np.int32)


def noise_tones_iq(seed, nsamples, chunk=1 << 22):
    """Receiver-like stream: approximately Gaussian noise (sum of four uniform
    bytes, sigma ~ 20 LSB around 128) plus two weak complex tones (period 8,
    amplitude 10 LSB; period 16 with a 3-sample step, amplitude 6 LSB), clipped
    to [0,255].  Dynamic range of the spectrum ~1e3 (configs C2-C5)."""
    out = np.empty(2 * nsamples, dtype=np.uint8)
    pos = 0
    while pos < nsamples:
        n = min(chunk, nsamples - pos)
        # one 64-bit word per complex sample: bytes 0-3 -> I noise, 4-7 -> Q noise
        b = splitmix64(seed, n, offset=pos).view(np.uint8).reshape(n, 8).astype(np.int32)
        ni = (b[:, 0:4].sum(axis=1) - 510) * 35 // 256
        nq = (b[:, 4:8].sum(axis=1) - 510) * 35 // 256
        k = np.arange(pos, pos + n, dtype=np.int64)
        ti = _TONE8_C[k % 8] + _TONE16_C[(3 * k) % 16]
        tq = _TONE8_S[k % 8] + _TONE16_S[(3 * k) % 16]
        out[2 * pos: 2 * (pos + n): 2] = np.clip(128 + ni + ti, 0, 255).astype(np.uint8)
        out[2 * pos + 1: 2 * (pos + n) + 1: 2] = np.clip(128 + nq + tq, 0, 255).astype(np.uint8)
        pos += n
    return out


def hann_window(N):
    """Periodic Hann window w[n] = 0.5 - 0.5 cos(2 pi n / N) as the float values
    the reference would read from a window file (acquisition.cxx:116)."""
    n = np.arange(N, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N)).astype(np.float32)
