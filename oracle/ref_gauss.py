"""
TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the reference's Gaussian model and
point sampler.  Never imported by the product path; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may use it.

Follows (file:line under /root/reference):
    gauss_handler.py:26-63    quaternion -> R, L = R diag(exp s), Sigma = L L^T
    gauss_handler.py:89-106   normals = R[:, argmin(log-scale)]
    gauss_handler.py:108-166  regularise (+5e-7 I), eigen check/clamp, cull mask
    gauss_handler.py:252-279  magnitudes (Knud-Thomsen ellipsoid area, p = 1.6075)
    gauss_to_pc.py:73-90      distribute_points (incl. the negative-slice quirk)
    gauss_to_pc.py:105-138    calculate_bin_sizes
    gauss_to_pc.py:140-275    MVN sampling + Mahalanobis rejection + first-k emission
    gauss_to_pc.py:277-371    generate_pointcloud bin loop and output order

Written against torch's CPU kernels (the reference is a torch program; its numerics are torch's
LAPACK/bmm) in float32 exactly where the reference is float32.  Pinned by tests/test_oracle_*.py
against tests/golden/*.npz, which oracle/make_golden.py produced by running the untouched
reference under oracle/ref_shim.py.

Noise is *injected*: `eps_fn(gids, attempt, n) -> float32 [len(gids), n, 3]`, keyed by the
Gaussian's index in the array handed to generate_pointcloud (see oracle/np_philox.py).
"""
from math import floor

import numpy as np
import torch

P_KT = 1.6075


# ----------------------------------------------------------------------------- geometry
def rotation_matrices(q: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:26-47 -- (r,x,y,z), no normalisation; result is float32."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.empty((q.shape[0], 3, 3), dtype=torch.float32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r * z)
    R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y)
    R[:, 2, 1] = 2 * (y * z + r * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def covariances(log_scales: torch.Tensor, q: torch.Tensor, modifier: float = 1.0) -> torch.Tensor:
    """gauss_handler.py:49-63."""
    R = rotation_matrices(q)
    D = torch.zeros((q.shape[0], 3, 3), dtype=torch.float32)
    s = modifier * log_scales
    for i in range(3):
        D[:, i, i] = torch.exp(s[:, i])
    L = R @ D
    return L @ L.transpose(1, 2)


def normals(log_scales: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:89-106: one-hot(argmin log-scale) rotated by R  == column argmin of R."""
    axis = torch.min(log_scales, 1)[1]
    onehot = torch.zeros((q.shape[0], 3), dtype=torch.float32)
    onehot[torch.arange(q.shape[0]), axis] = 1
    R = rotation_matrices(q)
    return torch.bmm(R, onehot.unsqueeze(2)).squeeze(2)


def strip_symmetric(cov: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:12-24 -- (xx, xy, xz, yy, yz, zz)."""
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2],
                        cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1).to(torch.float32)


def not_posdef(cov: torch.Tensor, eps: float) -> torch.Tensor:
    """gauss_handler.py:108-112."""
    return torch.any(torch.linalg.eigvals(cov).real <= eps, 1)


def validate_covariances(cov: torch.Tensor, regularise=True, eps=1e-7, min_eps=1e-8, iters=3):
    """gauss_handler.py:142-166.  Returns (validated covariances [all rows], keep mask)."""
    cov = cov.clone()
    if regularise:
        cov += 5e-7 * torch.eye(3, dtype=cov.dtype)            # :129-140
    for _ in range(iters):
        bad = not_posdef(cov, eps)
        if bad.sum() > 0:                                      # :114-127
            w, v = torch.linalg.eigh(cov[bad])
            w = torch.clamp(w, min=eps)
            cov[bad] = v @ torch.diag_embed(w) @ v.transpose(-1, -2)
    bad = not_posdef(cov, min_eps)
    return cov, ~bad


def magnitudes(cov: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:252-279; `weights` = contributions or opacities.  float64 result."""
    ev = torch.linalg.eigvals(cov).real
    a, b, c = torch.sqrt(ev[:, 0]), torch.sqrt(ev[:, 1]), torch.sqrt(ev[:, 2])
    radicand = (torch.pow(a * b, P_KT) + torch.pow(a * c, P_KT) + torch.pow(b * c, P_KT)) / 3.0
    area = 4.0 * torch.pi * torch.pow(radicand, 1.0 / P_KT)
    return (torch.sqrt(area) * weights).to(torch.float64)


# ----------------------------------------------------------------------------- allocation
def distribute_points(sizes: torch.Tensor, num_points: int) -> torch.Tensor:
    """gauss_to_pc.py:73-90 (float64 in, float64 out, round-half-even)."""
    total = torch.sum(sizes)
    ppg = torch.round(sizes * (num_points / total))
    zeros = torch.nonzero(ppg == 0).squeeze(1)
    k = int(min((num_points - ppg.sum()).item(), zeros.shape[0]))
    chosen = zeros[:k]          # k < 0 -> python negative slice: all but the last |k| (quirk kept)
    ppg[chosen] = 1
    return ppg


def calculate_bin_sizes(ppg_int: torch.Tensor):
    """gauss_to_pc.py:105-138."""
    dist = torch.bincount(ppg_int)
    dist = dist[dist != 0].numpy()
    g2 = np.absolute(np.gradient(np.gradient(dist)))
    bin_size = max(len(dist) // 100, 1)
    g2 = g2[:len(g2) - len(g2) % bin_size]
    sums = g2.reshape(-1, bin_size).sum(axis=1)
    cut = np.max(sums) // 50
    peak = int(np.argmax(sums))
    below = np.nonzero(sums[peak:] < cut)[0]
    start_bin = int(below[0]) if below.shape[0] != 0 else 1
    return start_bin, bin_size


def bin_table(ppg_int: torch.Tensor, exact: bool):
    """gauss_to_pc.py:308-337: list of (start, end, quota) per bin, in loop order.
    Quota n = floor(start + (end-start)/2) (NOT start: quirk kept); last bin end = start+1."""
    pd = torch.unique(ppg_int)
    if not exact:
        start_bin, bin_size = calculate_bin_sizes(ppg_int)
        tail = torch.mul(torch.unique(torch.ceil(pd[start_bin:] / bin_size)), bin_size)
        pd = torch.cat((pd[:start_bin], tail), 0)
    vals = [float(v) for v in pd]
    out = []
    for i, s in enumerate(vals):
        e = vals[i + 1] if i != len(vals) - 1 else s + 1
        out.append((s, e, floor(s + (e - s) / 2)))
    return out


# ----------------------------------------------------------------------------- sampling
def mahalanobis(means, samples, covs):
    """gauss_to_pc.py:92-103."""
    d = (means - samples).unsqueeze(2)
    m = torch.bmm(d.transpose(1, 2), torch.bmm(torch.inverse(covs), d))
    return torch.sqrt(m).reshape(-1)


def sample_bin(gids, means, covs, n, std, attempts, eps_fn):
    """gauss_to_pc.py:157-275 for one bin; returns (points [k,3] f32, owner [k] int64 = position
    in `gids` order, per-Gaussian emitted counts).  Emission = FIRST d draws, not the accepted."""
    G = means.shape[0]
    added = torch.zeros(G, dtype=torch.int64)
    pts, owner = [], []
    emitted = 0
    a = 0
    while emitted < n * G and a < attempts:
        act = torch.nonzero(added != n).squeeze(1)
        eps = torch.from_numpy(eps_fn(gids[act].numpy(), a, n))             # [Ga, n, 3]
        Lc = torch.linalg.cholesky(covs[act])                                # MultivariateNormal
        S = means[act].unsqueeze(1) + torch.matmul(Lc.unsqueeze(1), eps.unsqueeze(-1)).squeeze(-1)
        flat = S.reshape(-1, 3)
        d = mahalanobis(torch.repeat_interleave(means[act], n, dim=0), flat,
                        torch.repeat_interleave(covs[act], n, dim=0))
        acc = (d <= std).reshape(-1, n).sum(1)
        take = torch.minimum(n - added[act], acc)
        keep = torch.arange(n).unsqueeze(0) < take.unsqueeze(1)
        pts.append(S[keep])
        owner.append(torch.repeat_interleave(act, take))
        added[act] += take
        emitted += int(take.sum())
        a += 1
    if pts:
        return torch.cat(pts, 0), torch.cat(owner, 0), added
    return torch.zeros((0, 3)), torch.zeros((0,), dtype=torch.int64), added


def generate_pointcloud(xyz, covs, colours, normals_, weights, num_points, std=2.0, exact=False,
                        attempts=5, eps_fn=None, ppg=None):
    """gauss_to_pc.py:277-371.  Returns dict(points f32, colours, normals, gauss_index int64,
    ppg int32, bins)."""
    if ppg is None:          # `ppg` given: check the sampling alone on a fixed allocation
        sizes = magnitudes(covs, weights)
        ppg = distribute_points(sizes, num_points).to(torch.int32)
    bins = bin_table(ppg, exact)
    P, C, Nn, I = [], [], [], []
    for (s, e, n) in bins:
        idx = torch.where((ppg >= s) & (ppg < e))[0]
        if n <= 0 or idx.shape[0] < 1:
            continue
        P.append(xyz[idx]); C.append(colours[idx]); I.append(idx)
        if normals_ is not None:
            Nn.append(normals_[idx])
        if n <= 1:
            continue
        pts, owner, _ = sample_bin(idx, xyz[idx], covs[idx], n - 1, std, attempts, eps_fn)
        P.append(pts); C.append(colours[idx][owner]); I.append(idx[owner])
        if normals_ is not None:
            Nn.append(normals_[idx][owner])
    return dict(points=torch.cat(P, 0), colours=torch.cat(C, 0),
                normals=torch.cat(Nn, 0) if normals_ is not None else None,
                gauss_index=torch.cat(I, 0), ppg=ppg, bins=bins)
