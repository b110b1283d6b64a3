"""CPU: can torch.inverse's float32 3x3 arithmetic (the reference's mahalanobis(), gauss_to_pc.py:92-103) be reproduced bit for
bit?  200 000 covariances like the scene's; candidates: LU with partial pivoting + two triangular solves per column (with and
without fused multiply-adds) and the cofactor form the kernel uses.  Round 3 answer: no -- 0.6 % / 0.9 % / 0.2 % of the matrices
come out bit-identical (MKL's getrf / getri take another route); all agree to rounding, so an accept decision differs only for
a draw whose distance is within an ulp of the limit (tools/experiments/sampler_fuzz.py met one in 460 jobs)."""
import numpy as np, torch, itertools
torch.set_num_threads(1)
rng = np.random.default_rng(0)
n = 200000
# random SPD covariances like the scene's: R diag(s^2) R^T
q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
r, x, y, z = q.T
R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
              2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
              2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
s = rng.uniform(0.002, 0.02, size=(n, 3))
A = ((R * s[:, None, :]) @ np.transpose(R * s[:, None, :], (0, 2, 1))).astype(np.float32)
A = torch.from_numpy(A)
ref = torch.inverse(A).numpy()
f32 = np.float32
def fma(a, b, c):  # exact fused multiply-add in float32 via float64 (exact for f32 inputs: 24+24 bits product fits 53)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
def lu_inverse(A, use_fma, pivot=True, solve_order="cols"):
    a = A.numpy().copy()
    n = a.shape[0]
    perm = np.tile(np.arange(3), (n, 1))
    M = a.copy()
    rows = np.arange(n)
    for k in range(3):
        if pivot:
            p = k + np.argmax(np.abs(M[:, k:, k]), axis=1)
            sw = p != k
            tmp = M[rows, k].copy(); M[rows[sw], k] = M[rows[sw], p[sw]]; M[rows[sw], p[sw]] = tmp[sw]
            t2 = perm[rows, k].copy(); perm[rows[sw], k] = perm[rows[sw], p[sw]]; perm[rows[sw], p[sw]] = t2[sw]
        for i in range(k + 1, 3):
            l = (M[:, i, k] / M[:, k, k]).astype(f32)
            M[:, i, k] = l
            for j in range(k + 1, 3):
                M[:, i, j] = fma(-l, M[:, k, j], M[:, i, j]) if use_fma else (M[:, i, j] - (l * M[:, k, j]).astype(f32)).astype(f32)
    inv = np.zeros_like(a)
    for c in range(3):
        b = (perm == c).astype(f32)            # P e_c
        # forward: L y = b
        yv = b.copy()
        for i in range(1, 3):
            for j in range(i):
                yv[:, i] = fma(-M[:, i, j], yv[:, j], yv[:, i]) if use_fma else (yv[:, i] - (M[:, i, j] * yv[:, j]).astype(f32)).astype(f32)
        xv = yv.copy()
        for i in (2, 1, 0):
            for j in range(i + 1, 3):
                xv[:, i] = fma(-M[:, i, j], xv[:, j], xv[:, i]) if use_fma else (xv[:, i] - (M[:, i, j] * xv[:, j]).astype(f32)).astype(f32)
            xv[:, i] = (xv[:, i] / M[:, i, i]).astype(f32)
        inv[:, :, c] = xv
    return inv
for use_fma in (False, True):
    got = lu_inverse(A, use_fma)
    same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=(1, 2))
    print("LU pivot, fma", use_fma, "matrices bit-identical: %.4f" % same.mean(), "max rel diff %.2e" % (np.abs(got - ref) / np.abs(ref)).max())
# cofactor (the kernel's)
a = A.numpy()
a00,a01,a02,a10,a11,a12,a20,a21,a22 = [a[:, i, j] for i in range(3) for j in range(3)]
c00 = a11*a22 - a12*a21; c01 = a02*a21 - a01*a22; c02 = a01*a12 - a02*a11
c10 = a12*a20 - a10*a22; c11 = a00*a22 - a02*a20; c12 = a02*a10 - a00*a12
c20 = a10*a21 - a11*a20; c21 = a01*a20 - a00*a21; c22 = a00*a11 - a01*a10
det = a00*c00 + a01*c10 + a02*c20; idt = (f32(1.0)/det).astype(f32)
cof = np.stack([c00*idt, c01*idt, c02*idt, c10*idt, c11*idt, c12*idt, c20*idt, c21*idt, c22*idt], 1).reshape(-1,3,3)
print("cofactor vs torch: identical %.4f" % (cof.view(np.uint32) == ref.view(np.uint32)).all(axis=(1,2)).mean(), "max rel %.2e" % (np.abs(cof-ref)/np.abs(ref)).max())
