"""CPU (emulator): the blend kernel variants (g2pc_set_blend_variant 2 / 3 = two-wave, 4 / 5 = scalar-gather, 6 = dual-list
unroll 2) against the default dual-list kernel on random scenes, image sizes, tile limits and crowdings, default floor:
images, contributions at or above the floor and packed keys must be BIT-identical, colours too.
usage: python tools/experiments/blend_variant_fuzz.py <seed> <cases>.  Round 4: 40 cases x 5 variants, no difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3dgs-to-pc_amd'), os.path.join(ROOT, 'oracle')]
import numpy as np, torch
from g2pc import _native as nv
from emu_util import build_emu
nv._inject_for_tests(build_emu())
import gauss_render, camera_handler
from gauss_handler import Gaussians
from g2pc.synth import make_scene, make_cameras

rng = np.random.default_rng(int(sys.argv[1]))
FLOOR = 1e-6
bad = 0
for it in range(int(sys.argv[2])):
    W = int(rng.integers(40, 360)); H = int(rng.integers(30, 220)); n = int(rng.integers(200, 3000))
    crowd = float(rng.choice([0.2, 0.5, 1.0])); hi = float(rng.choice([0.02, 0.08, 0.3])); ncam = int(rng.integers(1, 4))
    mt = int(rng.choice([14, 25, 60]))
    sc = make_scene(n, 7000 + it, scale_lo=0.004, scale_hi=hi)
    xyz = sc.xyz * crowd
    tr, intr = make_cameras(ncam, width=W, height=H, focal=0.9 * W)
    out = {}
    t = time.time()
    for v in (1, 2, 3, 4, 5, 6):
        nv.experiments().g2pc_set_blend_variant(v)
        gauss_render.clear_context_pool()
        G = Gaussians(xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
        R.MAX_TILE_SIZE = mt
        imgs = [R(camera_handler.get_camera("python", torch.tensor(tr[k]), intr[k]))[0].numpy().copy() for k in tr]
        out[v] = (np.stack(imgs), R.gaussian_max_contribution.numpy().copy(), R.best_key.numpy().copy(), R.get_gaussian_colours().numpy().copy())
        del R
    nv.experiments().g2pc_set_blend_variant(1)
    base = out[1]
    above = base[1] >= FLOOR
    ok = True
    for v in (2, 3, 4, 5, 6):
        g = out[v]
        ok &= np.array_equal(base[0], g[0]) and np.array_equal(base[1][above], g[1][above]) and np.array_equal(base[2][above], g[2][above]) \
            and np.array_equal(base[3][above], g[3][above]) and float(np.abs(base[1] - g[1]).max()) <= FLOOR
    bad += (not ok)
    print(it, W, H, n, crowd, hi, ncam, mt, "OK" if ok else "MISMATCH", "%.1fs" % (time.time() - t), flush=True)
print("mismatches", bad)
