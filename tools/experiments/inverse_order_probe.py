"""CPU: can torch.inverse's float32 3x3 arithmetic (the reference's mahalanobis(), gauss_to_pc.py:92-103) be reproduced bit for
bit?  200 000 covariances like the scene's; candidates: LU with partial pivoting + two triangular solves per column (with and
without fused multiply-adds) and the cofactor form the kernel uses.  Round 3 answer: no -- 0.6 % / 0.9 % / 0.2 % of the matrices
come out bit-identical; the second half tries LAPACK's getrf + getri and getrf + getrs(identity) with reciprocal pivots and every
FMA placement (48 variants): at best 9 % identical (MKL takes another route); all agree to rounding, so an accept decision differs only for
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


# ---- second half: LAPACK-shaped restatements (getrf + getri, getrf + getrs on the identity), every FMA / reciprocal placement
import numpy as np, torch, itertools
torch.set_num_threads(1)
rng = np.random.default_rng(0)
n = 100000
q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
r, x, y, z = q.T
R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
              2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
s = rng.uniform(0.002, 0.02, size=(n, 3))
A = ((R * s[:, None, :]) @ np.transpose(R * s[:, None, :], (0, 2, 1))).astype(np.float32)
ref = torch.inverse(torch.from_numpy(A)).numpy()
f32 = np.float32
def mulsub(c, a, b, fma):   # c - a*b
    return (c.astype(np.float64) - a.astype(np.float64) * b.astype(np.float64)).astype(f32) if fma else (c - (a * b).astype(f32)).astype(f32)
def muladd(c, a, b, fma):
    return (c.astype(np.float64) + a.astype(np.float64) * b.astype(np.float64)).astype(f32) if fma else (c + (a * b).astype(f32)).astype(f32)
def getrf(A, fma, recip):
    M = A.copy(); N = M.shape[0]; rows = np.arange(N); piv = np.zeros((N, 3), np.int64)
    for k in range(3):
        p = k + np.argmax(np.abs(M[:, k:, k]), axis=1); piv[:, k] = p
        sw = p != k
        tmp = M[rows, k].copy(); M[rows[sw], k] = M[rows[sw], p[sw]]; M[rows[sw], p[sw]] = tmp[sw]
        if recip:
            rp = (f32(1.0) / M[:, k, k]).astype(f32)
            for i in range(k + 1, 3): M[:, i, k] = (M[:, i, k] * rp).astype(f32)
        else:
            for i in range(k + 1, 3): M[:, i, k] = (M[:, i, k] / M[:, k, k]).astype(f32)
        for i in range(k + 1, 3):
            for j in range(k + 1, 3):
                M[:, i, j] = mulsub(M[:, i, j], M[:, i, k], M[:, k, j], fma)
    return M, piv
def getri(M, piv, fma1, fma2):
    M = M.copy(); N = M.shape[0]; rows = np.arange(N)
    # strti2 upper non-unit
    for j in range(3):
        M[:, j, j] = (f32(1.0) / M[:, j, j]).astype(f32)
        ajj = -M[:, j, j]
        # x = U(0:j,0:j)^-1(already inverted part) * M[0:j, j]  (strmv upper, no-trans, non-unit)
        col = [M[:, i, j].copy() for i in range(j)]
        new = []
        for i in range(j):
            acc = (M[:, i, i] * col[i]).astype(f32)
            # strmv computes x := T x in place, for upper no-trans: for jj in 0..: temp=x[jj]; for ii<jj: x[ii]+=temp*T[ii][jj]; x[jj]*=T[jj][jj]
            new.append(None)
        xv = col[:]
        for jj in range(j):
            temp = xv[jj]
            for ii in range(jj):
                xv[ii] = muladd(xv[ii], temp, M[:, ii, jj], fma1)
            xv[jj] = (xv[jj] * M[:, jj, jj]).astype(f32)
        for i in range(j):
            M[:, i, j] = (ajj * xv[i]).astype(f32)
    # solve inv(A) * L = inv(U)
    for j in (1, 0):
        work = [M[:, i, j].copy() for i in range(j + 1, 3)]
        for i in range(j + 1, 3): M[:, i, j] = 0
        # A[:, j] -= A[:, j+1:] @ work   (sgemv: column-oriented axpy over k)
        for kk, i in enumerate(range(j + 1, 3)):
            for rrow in range(3):
                M[:, rrow, j] = mulsub(M[:, rrow, j], M[:, rrow, i], work[kk], fma2)
    for j in (1, 0):                      # column interchanges (j = n-2 .. 0)
        p = piv[:, j]; sw = p != j
        tmp = M[rows, :, j].copy(); M[rows[sw], :, j] = M[rows[sw], :, p[sw]]; M[rows[sw], :, p[sw]] = tmp[sw]
    return M
best = []
for fma0, recip, fma1, fma2 in itertools.product((False, True), repeat=4):
    M, piv = getrf(A, fma0, recip)
    got = getri(M, piv, fma1, fma2)
    same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=(1, 2)).mean()
    best.append((same, fma0, recip, fma1, fma2))
for b in sorted(best, reverse=True)[:6]:
    print("identical %.4f  getrf fma %s recip %s  trti2 fma %s  gemv fma %s" % b)
print("---- getrf + getrs(identity)")
def getrs(M, piv, fmaL, fmaU, udiv_recip):
    N = M.shape[0]; rows = np.arange(N)
    inv = np.zeros_like(M)
    # B = P I  (apply row interchanges to identity)
    B = np.tile(np.eye(3, dtype=f32), (N, 1, 1))
    for k in range(3):
        p = piv[:, k]; sw = p != k
        tmp = B[rows, k].copy(); B[rows[sw], k] = B[rows[sw], p[sw]]; B[rows[sw], p[sw]] = tmp[sw]
    # strsm left lower unit: for each rhs column, forward substitution (column-oriented axpy: for k: for i>k: b[i] -= b[k]*L[i][k])
    for c in range(3):
        b = [B[:, i, c].copy() for i in range(3)]
        for k in range(3):
            for i in range(k + 1, 3):
                b[i] = mulsub(b[i], b[k], M[:, i, k], fmaL)
        # strsm left upper non-unit: for k = 2..0: b[k] /= U[k][k]; for i<k: b[i] -= b[k]*U[i][k]
        for k in (2, 1, 0):
            b[k] = (b[k] * (f32(1.0) / M[:, k, k]).astype(f32)).astype(f32) if udiv_recip else (b[k] / M[:, k, k]).astype(f32)
            for i in range(k):
                b[i] = mulsub(b[i], b[k], M[:, i, k], fmaU)
        for i in range(3): inv[:, i, c] = b[i]
    return inv
best = []
for fma0, recip, fmaL, fmaU, ur in itertools.product((False, True), repeat=5):
    M, piv = getrf(A, fma0, recip)
    got = getrs(M, piv, fmaL, fmaU, ur)
    same = (got.view(np.uint32) == ref.view(np.uint32)).all(axis=(1, 2)).mean()
    best.append((same, fma0, recip, fmaL, fmaU, ur))
for b in sorted(best, reverse=True)[:6]:
    print("identical %.4f  getrf fma %s recip %s  L-solve fma %s  U-solve fma %s  U recip %s" % b)
