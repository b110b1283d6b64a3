"""
TEST INFRASTRUCTURE ONLY -- numpy restatement of the keyed noise used by the sampler.

The reference never seeds its RNG (torch's global Philox stream, gauss_to_pc.py:149), so
bit-parity of sampled points can only be defined for *injected* noise.  The product defines

    eps(seed, gid, attempt, k) in R^3   -- standard normal triple for draw k of Gaussian gid
                                           in sampling attempt `attempt`

as  Philox4x32-10(counter = (k, attempt, gid, 0), key = (seed_lo, seed_hi))  ->  4 x u32
    u_i = ((x_i >> 9) + 0.5) * 2^-23                 (exactly representable in fp32, in (0,1))
    r0 = sqrt(-2 ln u0), z0 = r0 cos(2 pi u1), z1 = r0 sin(2 pi u1)
    r1 = sqrt(-2 ln u2), z2 = r1 cos(2 pi u3)
where r and the trigonometric factors are the correctly rounded fp32 values of the real-valued functions of
the (exact) u_i -- evaluated here in fp64 and rounded; the HIP kernel evaluates them with fp32 logf / sqrtf /
sincospif(2u), each within ~1 ulp of the same real value -- and the products r*cos, r*sin are fp32 multiplies.  The parity harness feeds exactly these triples to the untouched reference by
patching torch.distributions' `_standard_normal` (see oracle/ref_shim.py), keyed by the global
Gaussian index so that one accept/reject flip cannot shift any other Gaussian's draws
(SURVEY.md §7 "RNG" / "Mahalanobis arithmetic").
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(x):
    return ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)


def keyed_normals(seed, gid, attempt, k):
    """eps[..., 3] float32 for broadcastable integer arrays gid / attempt / k."""
    gid = np.asarray(gid, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10(np.asarray(k, dtype=np.uint32),
                                   np.asarray(attempt, dtype=np.uint32),
                                   (gid & _MASK32).astype(np.uint32),
                                   (gid >> np.uint64(32)).astype(np.uint32),
                                   np.uint32(int(seed) & 0xFFFFFFFF),
                                   np.uint32((int(seed) >> 32) & 0xFFFFFFFF))
    u0, u1, u2, u3 = _u01(x0), _u01(x1), _u01(x2), _u01(x3)
    u0, u1, u2, u3 = (u.astype(np.float64) for u in (u0, u1, u2, u3))
    r0 = np.sqrt(-2.0 * np.log(u0)).astype(np.float32)
    r1 = np.sqrt(-2.0 * np.log(u2)).astype(np.float32)
    c0 = np.cos(2.0 * np.pi * u1).astype(np.float32)
    s0 = np.sin(2.0 * np.pi * u1).astype(np.float32)
    c1 = np.cos(2.0 * np.pi * u3).astype(np.float32)
    return np.stack([r0 * c0, r0 * s0, r1 * c1], axis=-1).astype(np.float32)
