"""Philox4x32-10 counter RNG (Salmon et al., SC'11), numpy restatement.

Test infrastructure (see oracle/__init__.py).  The HIP kernels in
image-generation_amd/csrc/philox.h implement the same function; the
known-answer vectors in tests/test_oracle_gibbs.py are the Random123 KATs.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)

# stream tags (4th counter word): which consumer of randomness
STREAM_GIBBS = 0
STREAM_INIT = 1
STREAM_GUMBEL = 2
STREAM_DROPOUT = 3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All args broadcastable uint32 arrays/scalars.

    Returns a tuple of four uint32 arrays.
    """
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, dtype=np.uint32),
        np.asarray(c1, dtype=np.uint32),
        np.asarray(c2, dtype=np.uint32),
        np.asarray(c3, dtype=np.uint32),
    )
    c0 = c0.astype(np.uint64)
    c1 = c1.astype(np.uint64)
    c2 = c2.astype(np.uint64)
    c3 = c3.astype(np.uint64)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0  # 64-bit product
            p1 = M1 * c2
            hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
            n0 = hi1 ^ c1 ^ np.uint64(k0)
            n1 = lo1
            n2 = hi0 ^ c3 ^ np.uint64(k1)
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return (
        c0.astype(np.uint32),
        c1.astype(np.uint32),
        c2.astype(np.uint32),
        c3.astype(np.uint32),
    )


def u32_to_unit_float(r):
    """Top 24 bits -> float32 in [0, 1): exact, identical on every platform."""
    return ((np.asarray(r, dtype=np.uint32) >> np.uint32(8)).astype(np.float32)) * np.float32(
        1.0 / 16777216.0
    )
