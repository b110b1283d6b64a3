"""How many (Gaussian, 128-pixel chunk) visits of the blend could be skipped because the Gaussian's alpha is below a
threshold on EVERY pixel of the chunk?  CPU, numpy/torch; bench scene (1 M Gaussians), camera 0, a sample of tiles."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), ROOT]
import numpy as np, torch
from oracle import ref_render as rr
from oracle import ref_gauss
from g2pc.synth import make_scene, make_cameras
torch.set_num_threads(8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = make_scene(n, 1237)
import gauss_handler
# covariances as the product builds them (R S S^T R^T)
s = torch.exp(sc.scales); q = sc.rots
r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
M = R * s[:, None, :]
cov3d = M @ M.transpose(1, 2)
tr, intr = make_cameras(50)
name = sorted(tr)[0]
cam = rr.get_camera(torch.tensor(tr[name]).float(), intr[name], colour_resolution=1280)
W, H = cam.image_width, cam.image_height
V = cam.world_view_transform
c2 = rr.cov2d(sc.xyz, cov3d, V, cam)
po = torch.cat([sc.xyz, torch.ones_like(sc.xyz[..., :1])], -1)
ph = po @ V @ cam.projection_matrix
pproj = ph * (1.0 / (ph[..., -1:] + 1e-6))
pview = po @ V
in_mask = pview[..., 2] <= -1e-6
ndc, depths, c2 = pproj[in_mask], pview[in_mask][:, 2], c2[in_mask]
op = sc.opacities[in_mask].float()
mx = ((ndc[..., 0] + 1) * W - 1.0) * 0.5
my = ((ndc[..., 1] + 1) * H - 1.0) * 0.5
det = c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] ** 2
mid = 0.5 * (c2[:, 0, 0] + c2[:, 1, 1])
l1 = mid + torch.sqrt((mid ** 2 - det).clip(min=0.1))
radii = 3.0 * torch.sqrt(l1).ceil()
rminx, rmaxx = (mx - radii).clip(0, W - 1.0), (mx + radii).clip(0, W - 1.0)
rminy, rmaxy = (my - radii).clip(0, H - 1.0), (my + radii).clip(0, H - 1.0)
conic = torch.linalg.inv(c2)
print("on-screen-ish Gaussians", int(in_mask.sum()), "median radius px", float(radii.median()), "p10/p90", float(radii.quantile(0.1)), float(radii.quantile(0.9)))
rng = np.random.default_rng(0)
tw, th = 40, 22
tot = {k: 0 for k in ("visits", "skip3e-8", "skip1e-6", "skip1e-5", "skip1e-4")}
for _ in range(24):
    x0 = int(rng.integers(0, W // tw)) * tw; y0 = int(rng.integers(0, 32)) * th
    w, h = tw, th
    m = (rmaxx.clip(max=x0 + w - 1) > rminx.clip(min=x0)) & (rmaxy.clip(max=y0 + h - 1) > rminy.clip(min=y0))
    idx = m.nonzero()[:, 0]
    if idx.numel() < 64: continue
    order = torch.argsort(depths[idx], descending=True)[:4000]
    idx = idx[order]
    ys, xs = torch.meshgrid(torch.arange(y0, y0 + h), torch.arange(x0, x0 + w), indexing="ij")
    dx = xs.reshape(-1, 1).float() - mx[idx][None]; dy = ys.reshape(-1, 1).float() - my[idx][None]
    cn = conic[idx]
    alpha = (torch.exp(-0.5 * (dx * dx * cn[:, 0, 0] + dy * dy * cn[:, 1, 1] + 2 * dx * dy * cn[:, 0, 1])) * op[idx][None]).clip(max=0.99)
    T = torch.cumprod(1 - alpha, 1)
    alpha = alpha.reshape(h, w, -1); T = T.reshape(h, w, -1)
    # chunks: horizontally adjacent pairs of 8x8 sub-blocks (16 x 8 pixels); the odd column pairs vertically -- here simply 16x8 windows
    for cy in range(0, h, 8):
        for cx in range(0, w, 16):
            a = alpha[cy:cy + 8, cx:cx + 16].reshape(-1, alpha.shape[2]); t = T[cy:cy + 8, cx:cx + 16].reshape(-1, alpha.shape[2])
            alive = (t > 1e-6).any(0)
            walk = int(alive.sum().item()) + 1
            walk = min(((walk + 63) // 64) * 64, a.shape[1])
            amax = a[:, :walk].max(0).values
            tot["visits"] += walk
            for k, thr in (("skip3e-8", 3e-8), ("skip1e-6", 1e-6), ("skip1e-5", 1e-5), ("skip1e-4", 1e-4)):
                tot[k] += int((amax < thr).sum())
print({k: v for k, v in tot.items()}, {k: round(v / tot["visits"], 3) for k, v in tot.items()})

# ---- tile-level: instances (Gaussian, tile) whose alpha stays below 2^-25 on the whole tile
tot_i = cul_i = 0
for _ in range(24):
    x0 = int(rng.integers(4, W // tw - 4)) * tw; y0 = int(rng.integers(4, 28)) * th
    m = (rmaxx.clip(max=x0 + tw - 1) > rminx.clip(min=x0)) & (rmaxy.clip(max=y0 + th - 1) > rminy.clip(min=y0))
    idx = m.nonzero()[:, 0]
    if idx.numel() < 64: continue
    idx = idx[torch.randperm(idx.numel())[:3000]]
    ys, xs = torch.meshgrid(torch.arange(y0, y0 + th), torch.arange(x0, x0 + tw), indexing="ij")
    dx = xs.reshape(-1, 1).float() - mx[idx][None]; dy = ys.reshape(-1, 1).float() - my[idx][None]
    cn = conic[idx]
    a = torch.exp(-0.5 * (dx * dx * cn[:, 0, 0] + dy * dy * cn[:, 1, 1] + 2 * dx * dy * cn[:, 0, 1])) * op[idx][None]
    amax = a.max(0).values
    tot_i += idx.numel(); cul_i += int((amax < 2.0 ** -25).sum())
print("tile-level: instances sampled", tot_i, "with alpha < 2^-25 on the whole tile:", cul_i, round(cul_i / max(tot_i, 1), 3))
