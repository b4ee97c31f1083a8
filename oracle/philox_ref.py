"""numpy restatement of libplship's counter-based normal stream (csrc/philox.h).  TEST INFRASTRUCTURE ONLY.

Philox4x32-10 is the published Random123 algorithm (Salmon et al., SC'11); the element -> counter mapping and
the Box-Muller pairing are this library's own definition (philox.h header) and have no reference counterpart:
the reference draws its noise from torch's CPU generator (src/samplers.py:30-35)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0 = np.uint64(k0) & MASK
    k1 = np.uint64(k1) & MASK
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c0, c1, c2, c3


def normal_matrix(rows: int, cols: int, seed: int, step: int, j_offset: int = 0) -> np.ndarray:
    """The (rows x cols) block [all rows, columns j_offset .. j_offset+cols) of the step's noise matrix."""
    i = np.arange(rows, dtype=np.uint64)[:, None]
    jg = (np.arange(cols, dtype=np.uint64) + np.uint64(j_offset))[None, :]
    ibase = i & ~np.uint64(4)
    shape = (rows, cols)
    x0, x1, x2, x3 = philox4x32_10(
        np.broadcast_to(ibase, shape),
        np.broadcast_to(jg, shape),
        np.full(shape, step & 0xFFFFFFFF, dtype=np.uint64),
        np.full(shape, (step >> 32) & 0xFFFFFFFF, dtype=np.uint64),
        seed & 0xFFFFFFFF,
        (seed >> 32) & 0xFFFFFFFF,
    )
    a = (x0 << np.uint64(32)) | x1
    b = (x2 << np.uint64(32)) | x3
    two_m53 = 2.0**-53
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) * two_m53
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) * two_m53
    rad = np.sqrt(-2.0 * np.log(u1))
    hi = np.broadcast_to((i & np.uint64(4)) != 0, shape)
    return np.where(hi, rad * np.sin(2.0 * np.pi * u2), rad * np.cos(2.0 * np.pi * u2))
