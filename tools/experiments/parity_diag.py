"""Diagnostics (GPU box): where do per-Gaussian colours differ from the reference at the benchmark's scale?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import camera_handler, gauss_render
from gauss_handler import Gaussians
from g2pc import _native as nv
from g2pc.synth import make_scene, make_cameras

g = np.load(os.path.join(ROOT, "tests", "golden", "render_py_cfg2_1m.npz"))
n = int(g["n"])
dev = torch.device("cuda:0")
sc = make_scene(n, int(g["seed"]))
tr, intr = make_cameras(50)
names = sorted(tr)
res = {}
for variant in (0, 1):
    for floor in (1e-6, 0.0):
        nv.experiments().g2pc_set_blend_variant(variant)
        G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
        R.t_floor = floor
        imgs = []
        for ci in g["cam_ids"]:
            cam = camera_handler.get_camera("python", torch.tensor(tr[names[int(ci)]]), intr[names[int(ci)]], colour_resolution=1280)
            imgs.append(R(cam)[0][::4, ::4].cpu().numpy())
        c = R.gaussian_max_contribution.cpu().numpy()
        cols = (R.get_gaussian_colours().cpu().numpy() / 255.0)[::16]
        refc, refcols = g["contrib_final"], g["colours_s16"] / 255.0
        d = np.abs(cols - refcols).max(axis=1)
        rc16 = refc[::16]
        bad = d > 1e-4
        out = {"image_max": float(max(np.abs(i - r).max() for i, r in zip(imgs, g["images_s4"]))),
               "image_frac": float(max((np.abs(i - r) > 1e-4).mean() for i, r in zip(imgs, g["images_s4"]))),
               "contrib_max": float(np.abs(c - refc).max()), "bad_total": int(bad.sum()), "n16": int(bad.size)}
        edges = [0, 1e-12, 1e-9, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 5e-2, 1.0]
        for lo, hi in zip(edges[:-1], edges[1:]):
            m = (rc16 > lo) & (rc16 <= hi)
            out["bin_%g_%g" % (lo, hi)] = [int(m.sum()), int((bad & m).sum()), int(((np.abs(cols).sum(axis=1) == 0) & m).sum())]
        out["ref_zero_colour_but_contrib"] = int(((np.abs(refcols).sum(axis=1) == 0) & (rc16 > 0)).sum())
        out["ours_zero_contrib"] = int((c[::16] == 0).sum()); out["ref_zero_contrib"] = int((rc16 == 0).sum())
        # are the bad ones explained by relative contribution differences (argmax flips)?
        rel = np.abs(c[::16] - rc16) / np.maximum(rc16, 1e-30)
        out["bad_rel_contrib_quantiles"] = [float(x) for x in np.quantile(rel[bad], [0.1, 0.5, 0.9])] if bad.any() else []
        out["good_rel_contrib_quantiles"] = [float(x) for x in np.quantile(rel[~bad & (rc16 > 0)], [0.1, 0.5, 0.9])]
        res["variant%d_floor%g" % (variant, floor)] = out
        R.close()
print(json.dumps(res, indent=1))
