"""CONTAINER ONLY (needs /root/reference) -- end to end against the UNTOUCHED reference: its own pipeline (python renderer over 1-3
cameras -> colours -> cull unrendered -> filter -> validate_covariances -> generate_pointcloud with keyed noise, run on the CPU
under oracle/ref_shim.py exactly as oracle/make_golden.py::gen_pipeline runs it) against the product's modules through the
emulator, on random small jobs (scene size, cameras, image size, point budget, binned / exact, Mahalanobis limit, attempts,
visibility threshold).  usage: python tools/experiments/pipeline_fuzz_vs_reference.py <seed> <cases>.  Round 3: 178 jobs (110 of them with min_opacity / bounding box / cull_large_percentage set) -- the
same culling mask, the same validate mask, the same number of points, xyz to 1.2e-7 and rgb (0..255) to 7.6e-5 in every one."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, '3dgs-to-pc_amd')]
import numpy as np, torch
from ref_shim import CudaToCpu, load_reference, reference_available
if not reference_available():
    sys.exit('the reference sources are not here (authoring container only)')
import make_golden as MG
ref = load_reference()
from g2pc import _native as nv
from emu_util import build_emu
nv._inject_for_tests(build_emu())
import gauss_render, camera_handler, gauss_to_pc as g2p
from gauss_handler import Gaussians
from g2pc.synth import make_scene, make_cameras
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    n = int(rng.integers(300, 3000)); ncam = int(rng.integers(1, 4)); W = int(rng.choice([160, 200, 320])); H = int(W * 9 // 16)
    num_points = int(n * rng.integers(1, 15)); exact = bool(rng.integers(0, 2)); std = float(rng.choice([1.0, 2.0])); attempts = int(rng.integers(1, 6))
    thr = float(rng.choice([0.0, 0.05, 0.2])); hi = float(rng.choice([0.02, 0.06])); seed = 8000 + it; noise_seed = int(rng.integers(0, 2**31))
    sc = make_scene(n, seed, scale_lo=0.004, scale_hi=hi)
    min_op = float(rng.choice([0.0, 0.2, 0.6])); cull_large = float(rng.choice([0.0, 0.0, 0.1]))
    bmin = [-0.8, -0.9, -1.0] if rng.random() < 0.3 else None; bmax = [0.9, 0.7, 0.8] if rng.random() < 0.3 else None
    transforms, intr = make_cameras(ncam, width=W, height=H, focal=0.9 * W)
    t0 = time.time()
    gh, gr, ch, rg2p = (ref[k] for k in ("gauss_handler", "gauss_render", "camera_handler", "gauss_to_pc"))
    with CudaToCpu():
        G = gh.Gaussians(sc.xyz.clone(), sc.scales.clone(), sc.rots.clone(), sc.colours.double(), sc.opacities.clone())
        G.calculate_normals()
        R = gr.get_renderer("python", G.xyz, torch.unsqueeze(torch.clone(G.opacities), 1), G.colours, G.covariances, visible_gaussian_threshold=thr)
        for name in transforms:
            cam = ch.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=None)
            with MG.stable_depth_ties():
                R(cam)
        G.colours = R.get_gaussian_colours()
        G.add_gaussians_to_cull(R.get_visible_gaussians())
        G.apply_min_opacity(min_op); G.apply_bounding_box(bmin, bmax)
        try:
            G.cull_large_gaussians(cull_large); r_cl_err = None
        except Exception as e:
            r_cl_err = type(e).__name__
        r_culled = G.filter_gaussians()
        r_contrib = R.get_total_gaussian_contributions()[r_culled]
        r_keep = G.validate_covariances()
        r_contrib = r_contrib[r_keep]
        try:
            with MG.KeyedNoise(rg2p, G.xyz, noise_seed):
                r_pts, r_cols, r_nrms = rg2p.generate_pointcloud(G, num_points, exact_num_points=exact, mahalanobis_distance_std=std, calculate_normals=True,
                                                                 num_sample_attempts=attempts, contributions=r_contrib, device="cpu", quiet=True)
            r_err = None
        except Exception as e:
            r_err = type(e).__name__
    t_ref = time.time() - t0
    # ---- product
    gauss_render.clear_context_pool()
    P = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    P.calculate_normals()
    PR = gauss_render.get_renderer("python", P.xyz, torch.unsqueeze(torch.clone(P.opacities), 1), P.colours, P.covariances, visible_gaussian_threshold=thr)
    PR.t_floor = 0.0
    for name in transforms:
        PR(camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=None))
    P.colours = PR.get_gaussian_colours()
    P.add_gaussians_to_cull(PR.get_visible_gaussians())
    P.apply_min_opacity(min_op); P.apply_bounding_box(bmin, bmax)
    if r_cl_err is None:
        P.cull_large_gaussians(cull_large)
    culled = P.filter_gaussians()
    contrib = PR.get_total_gaussian_contributions()[culled]
    keep = P.validate_covariances()
    contrib = contrib[keep]
    try:
        pts, cols, nrms = g2p.generate_pointcloud(P, num_points, exact_num_points=exact, mahalanobis_distance_std=std, calculate_normals=True,
                                                  num_sample_attempts=attempts, contributions=contrib, device="cpu", quiet=True, seed=noise_seed)
        p_err = None
    except Exception as e:
        p_err = type(e).__name__
    tag = "op %.1f cl %.1f bb %s %s n %d cams %d %dx%d pts %d exact %s std %.0f att %d thr %.2f" % (min_op, cull_large, bmin is not None, bmax is not None, n, ncam, W, H, num_points, exact, std, attempts, thr)
    if r_err or p_err:
        ok = (r_err is not None) and (p_err is not None)
        bad += (not ok); print(it, tag, "raised ref", r_err, "product", p_err, "OK" if ok else "MISMATCH", flush=True); continue
    m_c = np.array_equal(culled.numpy(), MG._np(r_culled)); m_k = np.array_equal(keep.numpy(), MG._np(r_keep))
    dc = float(np.abs(contrib.numpy() - MG._np(r_contrib)).max()) if m_c and m_k and contrib.numel() else (0.0 if m_c and m_k else -1)
    same_rows = pts.shape[0] == r_pts.shape[0]
    dp = float(np.abs(pts.numpy() - MG._np(r_pts)).max()) if same_rows and pts.shape[0] else (0.0 if same_rows else -1)
    dcol = float(np.abs(cols.numpy() - MG._np(r_cols).astype(np.float32)).max()) if same_rows and pts.shape[0] else 0.0
    ok = m_c and m_k and 0 <= dc < 1e-4 and same_rows and 0 <= dp < 1e-4
    bad += (not ok)
    print(it, tag, "| culled", m_c, "keep", m_k, "contrib %.1e rows %d/%d xyz %.1e rgb %.1e" % (dc, pts.shape[0], r_pts.shape[0], dp, dcol), "OK" if ok else "MISMATCH", "ref %.1fs total %.1fs" % (t_ref, time.time() - t0), flush=True)
print("mismatches", bad)
