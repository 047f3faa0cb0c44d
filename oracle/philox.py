"""ORACLE (test infrastructure) -- the build's counter-based Gaussian stream, restated in numpy.

The reference draws noise with `torch.randn_like(batch, device='cuda')` (smoothing.py:96): a
device-global Philox stream that is neither reproducible on ROCm nor independent of the batch size.
The build replaces it by a stateless stream:

    element e (flat CHW index) of Monte-Carlo sample s under seed S
      g = e // 4, lane = e % 4
      (r0,r1,r2,r3) = Philox4x32-10(counter = (g, 0, s_lo, s_hi), key = (S_lo, S_hi))
      u1 = ((r[2p]   >> 8) + 1) * 2^-24   in (0,1]        p = lane // 2
      u2 =  (r[2p+1] >> 8)      * 2^-24   in [0,1)
      z  = sqrt(-2 ln u1) * (cos(2 pi u2) if lane even else sin(2 pi u2))

so counts are identical for any batch size and any sharding of the sample range over GPUs.
certifiedgpt_amd/csrc/philox.h is the product implementation (fp32 arithmetic); this file evaluates the
same definition in float64 and rounds to float32, so the two agree to a few fp32 ulps, not bit-for-bit.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11).  Inputs broadcastable uint32-valued arrays."""
    c0 = np.asarray(c0, dtype=np.uint64); c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64); c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def _normals_from_words(r0, r1, r2, r3):
    """4 uint32 words -> 4 float64 normals (Box-Muller, definition in the module docstring)."""
    out = []
    for ra, rb in ((r0, r1), (r2, r3)):
        u1 = ((ra >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
        u2 = (rb >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
        rad = np.sqrt(-2.0 * np.log(u1))
        out.append(rad * np.cos(2.0 * np.pi * u2))
        out.append(rad * np.sin(2.0 * np.pi * u2))
    return out


def normal_stream(seed: int, stream: int, numel: int, hi_word: int = 0) -> np.ndarray:
    """numel N(0,1) draws for (seed, stream): element e uses counter (e//4, hi_word, stream_lo, stream_hi)."""
    groups = (numel + 3) // 4
    g = np.arange(groups, dtype=np.uint64)
    r = philox4x32_10(g, np.uint64(hi_word), np.uint64(stream & 0xFFFFFFFF), np.uint64((stream >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z = np.stack(_normals_from_words(*r), axis=1).reshape(-1)[:numel]
    return z.astype(np.float32)


def noise_batch(seed: int, first_sample: int, num: int, shape) -> np.ndarray:
    """eps[b] for samples first_sample .. first_sample+num-1, each of `shape` (C,H,W); float32 N(0,1)."""
    numel = int(np.prod(shape))
    return np.stack([normal_stream(seed, first_sample + b, numel).reshape(shape) for b in range(num)], axis=0)
