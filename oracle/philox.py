"""numpy restatement of the on-device random streams (test infrastructure only).

The reference draws latents with ``torch.randn`` and uniforms with a numpy
``Generator`` (``src/nessai/flows/distributions.py:58-65``,
``src/nessai/proposal/flowproposal/flowproposal.py:493``); neither stream can be
reproduced bit-for-bit on a GPU, so the build defines its own counter-based
streams (Philox4x32-10, Salmon et al. 2011) and pins them here:

* latents:  counter = (row lo, row hi, quad, 0), key = seed; 4 outputs ->
  2 Box-Muller pairs -> columns 4*quad .. 4*quad+3 of the row;
* uniforms: counter = (row lo, row hi, 0, 1); u = 53-bit fraction of outputs 0, 1.
"""

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; all inputs uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(x):
    return (x.astype(np.float32) * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)).astype(np.float32)


def _box_muller(a, b):
    u1, u2 = _u01(a), _u01(b)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0) * u2).astype(np.float64) * np.pi
    return (r * np.cos(ang).astype(np.float32)).astype(np.float32), (r * np.sin(ang).astype(np.float32)).astype(np.float32)


def latents(seed, row_offset, n, dims):
    """float32 ``(n, dims)`` latents exactly as ``nsa_flow_kernel`` draws them."""
    rows = np.uint64(row_offset) + np.arange(n, dtype=np.uint64)
    lo = (rows & MASK).astype(np.uint32)
    hi = (rows >> np.uint64(32)).astype(np.uint32)
    k0 = np.uint32(np.uint64(seed) & MASK)
    k1 = np.uint32(np.uint64(seed) >> np.uint64(32))
    out = np.zeros((n, 4 * ((dims + 3) // 4)), dtype=np.float32)
    for q in range((dims + 3) // 4):
        r0, r1, r2, r3 = philox4x32_10(lo, hi, np.uint32(q), np.uint32(0), k0, k1)
        z0, z1 = _box_muller(r0, r1)
        z2, z3 = _box_muller(r2, r3)
        out[:, 4 * q + 0] = z0
        out[:, 4 * q + 1] = z1
        out[:, 4 * q + 2] = z2
        out[:, 4 * q + 3] = z3
    return out[:, :dims]


def uniforms(seed, row_offset, n):
    """float64 ``(n,)`` uniforms in [0, 1) exactly as ``nsa_accept_flag_kernel`` draws them."""
    rows = np.uint64(row_offset) + np.arange(n, dtype=np.uint64)
    lo = (rows & MASK).astype(np.uint32)
    hi = (rows >> np.uint64(32)).astype(np.uint32)
    k0 = np.uint32(np.uint64(seed) & MASK)
    k1 = np.uint32(np.uint64(seed) >> np.uint64(32))
    r0, r1, _, _ = philox4x32_10(lo, hi, np.uint32(0), np.uint32(1), k0, k1)
    frac = ((r0 >> np.uint32(5)).astype(np.uint64) << np.uint64(26)) | (r1 >> np.uint32(6)).astype(np.uint64)
    return frac.astype(np.float64) * (1.0 / 9007199254740992.0)
