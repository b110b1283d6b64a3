"""CONTAINER ONLY (needs /root/reference) -- the python-renderer ORACLE (oracle/ref_render.py) against the untouched reference, bit for
bit, on random scenes, image sizes and memory pins (ref_shim.TILE_PIN -> max_gaussians_per_tile = pin, max_tile_size = pin // 1000,
gauss_render.py:440-444): crowded scenes under small pins make the reference's queue split leaves by count.
usage: python tools/experiments/oracle_vs_reference_fuzz.py <seed> <cases>.  Round 3: 25 cases, images / contributions / colours
bit-equal in every one (the fixtures tests/test_oracle_render.py holds the oracle to are five more)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, '3dgs-to-pc_amd')]
import numpy as np, torch
import ref_shim
from ref_shim import CudaToCpu, load_reference, reference_available
if not reference_available():
    sys.exit('the reference sources are not here (authoring container only)')
import make_golden as MG
ref = load_reference()
import ref_gauss as RG, ref_render as RR
from g2pc.synth import make_scene, make_cameras
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
gh, gr, ch = ref["gauss_handler"], ref["gauss_render"], ref["camera_handler"]
for it in range(int(sys.argv[2])):
    pin = int(rng.choice([6000, 10000, 25000, 60000])); n = int(rng.integers(2000, 40000)); crowd = float(rng.choice([0.05, 0.15, 0.4, 1.0]))
    W = int(rng.integers(40, 260)); H = int(rng.integers(30, 160)); ncam = int(rng.integers(1, 3)); hi = float(rng.choice([0.012, 0.04]))
    sc = make_scene(n, 9000 + it, scale_lo=0.003, scale_hi=hi)
    xyz = sc.xyz * crowd
    tr, intr = make_cameras(ncam, width=W, height=H, focal=0.9 * W)
    saved = ref_shim.TILE_PIN; ref_shim.TILE_PIN = pin
    t = time.time()
    try:
        with CudaToCpu():
            G = gh.Gaussians(xyz.clone(), sc.scales.clone(), sc.rots.clone(), sc.colours.double(), sc.opacities.clone())
            R = gr.get_renderer("python", G.xyz, torch.unsqueeze(torch.clone(G.opacities), 1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
            cov = RG.covariances(sc.scales, sc.rots)
            O = RR.PythonRendererOracle(xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05, max_tile_size=pin // 1000, max_gaussians_per_tile=pin)
            ok = bool(torch.equal(G.covariances, cov))
            for name in tr:
                cam = ch.get_camera("python", torch.tensor(tr[name]), intr[name], colour_resolution=None)
                with MG.stable_depth_ties():
                    img = R(cam)[0]
                oimg = O(RR.get_camera(torch.tensor(tr[name]), intr[name]))
                ok = ok and bool(torch.equal(img.float(), oimg.float()))
            ok = ok and bool(torch.equal(R.gaussian_max_contribution.reshape(-1), O.max_contribution.reshape(-1))) and bool(torch.equal(R.get_gaussian_colours(), O.get_gaussian_colours()))
    finally:
        ref_shim.TILE_PIN = saved
    bad += (not ok)
    print(it, "pin", pin, "n", n, "crowd", crowd, W, H, "cams", ncam, "BIT-EQUAL" if ok else "MISMATCH", "%.1fs" % (time.time() - t), flush=True)
print("mismatches", bad)
