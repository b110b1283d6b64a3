"""Debug aid (GPU box): per-pixel comparison of the per-tile colour buffer of one camera rendered by the two-call path and by
the batched camera call with the work hand-over forced after `SPLIT` batches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd")]
import numpy as np, torch
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras

dev = "cuda:0"
SPLIT = int(os.environ.get("SPLIT", "1"))
sc = make_scene(120_000, 11, device=dev, scale_lo=0.004, scale_hi=0.03)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
tr, intr = make_cameras(6, width=640, height=360, focal=550.0)
names = sorted(tr)
cams = [camera_handler.get_camera("python", torch.tensor(tr[n]), intr[n]) for n in names]


def tune(lpt, split, min_left):
    L = nv.lib()
    for i, v in ((0, lpt), (1, split), (2, min_left)):
        nv.check(L.g2pc_set_blend_tuning(i, v), "tune")


def render(pipelined, batch, which):
    gauss_render.clear_context_pool()
    gauss_render.CAMERA_BATCH = batch
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
    R(cams[0], return_image=not pipelined)              # (the pipeline's first camera always takes the two-call path)
    bufs = []
    for k in which:
        R(cams[k], return_image=not pipelined)
        if not pipelined:
            torch.cuda.synchronize()
            bufs.append(R.sync_scratch.tilebuf.clone())
    if pipelined:
        R.flush()
        torch.cuda.synchronize()
        bufs = [t.clone() for t in R.ctx.cam_tilebufs[:len(which)]]
    keys = R.best_key.clone()
    lay = R._layout(640, 360)
    R.close()
    return bufs, keys, lay


tune(1, 0, 1)
ref, kref, lay = render(False, 1, [1, 2])
H = lay.host
nx = H["nx"]
off = H["tile_pix_off"]
for batch in (1, 2):
    for rep in range(2):
        tune(1, SPLIT, 1)
        got, kgot, _ = render(True, batch, [1, 2])
        for ci, (a, b) in enumerate(zip(ref, got)):
            n = min(a.numel(), b.numel())
            d = (a[:n] - b[:n]).abs().reshape(-1, 3).max(1).values.cpu().numpy()
            bad = np.nonzero(d > 1e-4)[0]
            print("batch", batch, "rep", rep, "camera", ci + 1, "pixels", d.shape[0], "bad", bad.shape[0], "max", float(d.max()),
                  "keys equal", bool(torch.equal(kref, kgot)))
            if bad.shape[0]:
                t = np.searchsorted(off, bad, side="right") - 1
                p = bad - off[t]
                w = H["ws"][t % nx]
                x, y = p % w, p // w
                sbx, sby = x // 8, y // 8
                quarter = (y % 8) // 2
                for name, v in (("tile", t), ("x%8", x % 8), ("y%8", y % 8), ("quarter", quarter), ("sb_x", sbx), ("sb_y", sby)):
                    u, c = np.unique(v, return_counts=True)
                    print("   ", name, dict(zip(u.tolist()[:24], c.tolist()[:24])))
                av, bv = a[:n].reshape(-1, 3).cpu().numpy(), b[:n].reshape(-1, 3).cpu().numpy()
                for i in bad[:8]:
                    print("    pixel", int(i), "ref", av[i], "got", bv[i])
