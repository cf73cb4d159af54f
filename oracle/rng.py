"""Counter-based RNG shared (by specification, not by code) with the HIP library.

Philox4x32-10 (Salmon et al., SC'11; the published constants).  Draw ``i`` of stream
``(seed, stream)`` is word ``i & 3`` of ``philox(counter=(i>>2, 0, stream, 0), key=(seed_lo,
seed_hi))``; a keep-mask element is 1 iff ``word * 2**-32 < keep_prob`` evaluated in float32
exactly as the kernel does (``(float)(word >> 8) * 2**-24 < keep``).  The reference itself uses
TensorFlow's stateful ``random_uniform`` (Modules.py:41-45, ZoneoutLSTMCell.py:266-271), which
cannot be reproduced; parity is defined on identical injected masks.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; all inputs uint32 arrays (broadcastable)."""
    c0 = np.asarray(c0, np.uint32); c1 = np.asarray(c1, np.uint32)
    c2 = np.asarray(c2, np.uint32); c3 = np.asarray(c3, np.uint32)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform_words(n, seed, stream):
    """First ``n`` uint32 draws of stream (seed, stream)."""
    n = int(n)
    nblk = (n + 3) // 4
    idx = np.arange(nblk, dtype=np.uint64)
    c0 = (idx & MASK32).astype(np.uint32)
    c1 = (idx >> np.uint64(32)).astype(np.uint32)
    c2 = np.full(nblk, np.uint32(stream & 0xFFFFFFFF), np.uint32)
    c3 = np.zeros(nblk, np.uint32)
    r = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(r, axis=1).reshape(-1)[:n]


def keep_mask(shape, seed, stream, keep_prob):
    """uint8 Bernoulli(keep_prob) mask of ``shape`` (row-major element order)."""
    n = int(np.prod(shape))
    w = uniform_words(n, seed, stream)
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u < np.float32(keep_prob)).astype(np.uint8).reshape(shape)
