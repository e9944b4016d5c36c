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


def noise_tones_iq(seed, nsamples, chunk=1 << 22, first=0):
    """Receiver-like stream: approximately Gaussian noise (sum of four uniform
    bytes, sigma ~ 20 LSB around 128) plus two weak complex tones (period 8,
    amplitude 10 LSB; period 16 with a 3-sample step, amplitude 6 LSB), clipped
    to [0,255].  Dynamic range of the spectrum ~1e3 (configs C2-C5).
    `first`: index of the first complex sample (a shard of the seed's stream)."""
    out = np.empty(2 * nsamples, dtype=np.uint8)
    pos = 0
    while pos < nsamples:
        n = min(chunk, nsamples - pos)
        # one 64-bit word per complex sample: bytes 0-3 -> I noise, 4-7 -> Q noise
        b = splitmix64(seed, n, offset=first + pos).view(np.uint8).reshape(n, 8).astype(np.int32)
        ni = (b[:, 0:4].sum(axis=1) - 510) * 35 // 256
        nq = (b[:, 4:8].sum(axis=1) - 510) * 35 // 256
        k = np.arange(first + pos, first + pos + n, dtype=np.int64)
        ti = _TONE8_C[k % 8] + _TONE16_C[(3 * k) % 16]
        tq = _TONE8_S[k % 8] + _TONE16_S[(3 * k) % 16]
        out[2 * pos: 2 * (pos + n): 2] = np.clip(128 + ni + ti, 0, 255).astype(np.uint8)
        out[2 * pos + 1: 2 * (pos + n) + 1: 2] = np.clip(128 + nq + tq, 0, 255).astype(np.uint8)
        pos += n
    return out


def _wrap64(v):
    """A 64-bit pattern as the signed Python int torch's int64 holds."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def noise_tones_iq_torch(seed, nsamples, device, first=0, chunk=1 << 24):
    """The same bytes as noise_tones_iq(seed, nsamples, first=first), generated on `device`
    with torch integer arithmetic (int64 with two's-complement wrap-around = the uint64
    arithmetic of splitmix64; logical right shifts are arithmetic shifts masked).  Used
    where a 0.5 GB stream would take a minute of numpy: bench.py and the full-size tests
    (which check it against the committed stream checksums)."""
    import torch
    gamma, m1, m2 = (_wrap64(int(c)) for c in (_GAMMA, _M1, _M2))

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    t8c = torch.tensor(_TONE8_C, dtype=torch.int32, device=device)
    t8s = torch.tensor(_TONE8_S, dtype=torch.int32, device=device)
    t16c = torch.tensor(_TONE16_C, dtype=torch.int32, device=device)
    t16s = torch.tensor(_TONE16_S, dtype=torch.int32, device=device)
    out = torch.empty(2 * nsamples, dtype=torch.uint8, device=device)
    pos = 0
    while pos < nsamples:
        n = min(chunk, nsamples - pos)
        k = torch.arange(first + pos, first + pos + n, dtype=torch.int64, device=device)
        z = (k + 1) * gamma + _wrap64(int(seed))
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        z = z ^ lsr(z, 31)
        si = ((z & 0xff) + (lsr(z, 8) & 0xff) + (lsr(z, 16) & 0xff) + (lsr(z, 24) & 0xff)).to(torch.int32)
        sq = ((lsr(z, 32) & 0xff) + (lsr(z, 40) & 0xff) + (lsr(z, 48) & 0xff) + (lsr(z, 56) & 0xff)).to(torch.int32)
        ni = ((si - 510) * 35) >> 8                   # floor division by 256
        nq = ((sq - 510) * 35) >> 8
        k8, k16 = (k % 8), ((3 * k) % 16)
        vi = 128 + ni + t8c[k8] + t16c[k16]
        vq = 128 + nq + t8s[k8] + t16s[k16]
        out[2 * pos: 2 * (pos + n): 2] = vi.clamp_(0, 255).to(torch.uint8)
        out[2 * pos + 1: 2 * (pos + n) + 1: 2] = vq.clamp_(0, 255).to(torch.uint8)
        pos += n
    return out


def hann_window(N):
    """Periodic Hann window w[n] = 0.5 - 0.5 cos(2 pi n / N) as the float values
    the reference would read from a window file (acquisition.cxx:116)."""
    n = np.arange(N, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N)).astype(np.float32)
