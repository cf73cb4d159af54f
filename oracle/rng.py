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


def keep_mask_rows(shape, batch_axis, seed, stream, sample0, keep_prob):
    """Sample-keyed keep-mask of a 3-D ``shape`` whose ``batch_axis`` (0: [B, T, C], 1: [S, B, C]) indexes the samples of a
    batch: the mask of sample ``b`` is drawn from its own Philox stream, counter = (block, sample0 + b, stream, 0), element
    (o, c) of the sample's [outer, inner] slice being draw ``o * inner + c`` of it.  ``sample0`` = the global index of this
    rank's first sample, so a sample sees the same mask however the global batch is sharded."""
    if batch_axis == 0:
        B, outer, inner = int(shape[0]), 1, int(np.prod(shape[1:]))
    else:
        outer, B, inner = int(shape[0]), int(shape[1]), int(np.prod(shape[2:]))
    n = outer * inner
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint32)[None, :].repeat(B, 0)
    smp = ((int(sample0) + np.arange(B, dtype=np.uint64)) & MASK32).astype(np.uint32)[:, None].repeat(nblk, 1)
    c2 = np.full((B, nblk), np.uint32(stream & 0xFFFFFFFF), np.uint32)
    r = philox4x32_10(blk, smp, c2, np.zeros((B, nblk), np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(r, axis=2).reshape(B, -1)[:, :n]
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    m = (u < np.float32(keep_prob)).astype(np.uint8)                      # [B, outer * inner]
    if batch_axis == 0:
        return m.reshape(shape)
    return np.ascontiguousarray(m.reshape(B, outer, inner).transpose(1, 0, 2)).reshape(shape)
