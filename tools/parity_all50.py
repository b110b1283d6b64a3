"""
CHECKER (test infrastructure): the BENCHMARKED job itself -- BASELINE configs[2], ALL 50 cameras, through the PRODUCTION path --
against the untouched reference.

tests/golden/render_py_cfg2_1m_all50.npz is the output of /root/reference (python renderer under the stable tie rule, then the
reference's own conversion tail; oracle/make_golden.py render_all, ~1-3 h of container CPU) on the bench scene with every camera
of the 50-camera rig in the order convert_3dgs_to_pc walks them (gauss_to_pc.py:437-454).  `run(device)` sends the same job
through `gauss_to_pc.convert_gaussians_to_pc` exactly as bench.py's timed loop does -- pipelined cameras (PIPELINE_STREAMS
streams x CAMERA_BATCH-camera graph replays, deferred colour resolve, pooled context), visible cull, filter, validate,
magnitudes, distribute, sampler -- with two hooks that change no arithmetic: the cameras carry the REFERENCE's matrices
(host arithmetic whose last bits depend on the host's BLAS: the fixture holds them, as for tools/parity_cfg2.py) and the
renderer object is kept so that its final state can be read.  Gates:

  mask_flips                 visible mask (running max > 0.05) over all 1 M Gaussians after 50 cameras
  contrib_max                final running-max contribution, every 4th Gaussian
  winner_camera_mismatch     Gaussians whose running maximum was set by another camera than in the reference (the slot field of
                             the packed keys against the camera that last raised the reference's maximum): cross-camera order,
                             strict-> ties (earliest camera wins, gauss_render.py:387-395), deferred resolve across streams
  colour_max                 per-Gaussian colour (colour of the arg-max pixel IN THE WINNING CAMERA), every 16th Gaussian
  culled_equal / keep_equal  the kept index set
  ppg_*                      point quotas; every end-to-end difference is EXPLAINED or counted as unexplained: a quota may differ
                             only by one and only where the reference's unrounded quota lies within the rounding distance that
                             the contribution difference of that very Gaussian (and of the sum) can move it (ppg_flips_explained)
  sample_*                   the 10 M-point cloud the job returned against every 256th row of the reference's

Used by tests/test_gpu_parity_scale.py and bench.py's `parity` block.  Never imported by the product package.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

from parity_cfg2 import GOLDEN, _bits, k1_hash8, match_rows, explain_quota_flips  # noqa: E402


def available(tag="1m_all50"):
    return os.path.isfile(os.path.join(GOLDEN, "render_py_cfg2_%s.npz" % tag))


def run(device="cuda:0", tag="1m_all50", sampler=True, t_floor=None):
    import camera_handler
    import gauss_render
    import gauss_to_pc as g2p
    from gauss_handler import Gaussians
    from g2pc import ops
    from g2pc.synth import make_scene, make_cameras

    t_start = time.perf_counter()
    g = np.load(os.path.join(GOLDEN, "render_py_cfg2_%s.npz" % tag))
    n, seed, rig = int(g["n"]), int(g["seed"]), int(g["rig"])
    width, height, focal = int(g["width"]), int(g["height"]), float(g["focal"])
    num_points = int(g["num_points"])
    dev = torch.device(device)
    sc = make_scene(n, seed)
    transforms, intr = make_cameras(rig, width=width, height=height, focal=focal)
    names = sorted(transforms)
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    out = {"gaussians": n, "cameras": rig, "resolution": "%dx%d" % (width, height), "path": "convert_gaussians_to_pc (production)",
           "oracle": "untouched reference on CPU, all %d cameras (oracle/make_golden.py render_all)" % rig,
           "reference_tie_rule": str(g["tie_rule"])}
    c9 = G.covariances.reshape(n, 9).cpu().numpy()
    out["cov3d_rows_differing"] = int((k1_hash8(*[c9[:, j] for j in (0, 1, 2, 4, 5, 8)]) != g["cov3d_hash8"]).sum())

    # hook 1: the reference's camera matrices (the rig is built in instalments by get_cameras; names -> fixture rows)
    index = {nm: i for i, nm in enumerate(names)}
    orig_get_cameras, orig_get_renderer = camera_handler.get_cameras, gauss_render.get_renderer
    bits = [0]

    def get_cameras(renderer_type, tr, intrinsics, **kw):
        cams = orig_get_cameras(renderer_type, tr, intrinsics, **kw)
        for nm, cam in cams.items():
            i = index[nm]
            view, proj = torch.from_numpy(g["cam_view"][i]), torch.from_numpy(g["cam_proj"][i])
            assert torch.allclose(cam.world_view_transform.cpu(), view, rtol=1e-5, atol=1e-6), nm
            assert torch.allclose(cam.projection_matrix.cpu(), proj, rtol=1e-5, atol=1e-6), nm
            bits[0] += int((cam.world_view_transform.cpu().numpy().view(np.uint32) != g["cam_view"][i].view(np.uint32)).sum() +
                           (cam.projection_matrix.cpu().numpy().view(np.uint32) != g["cam_proj"][i].view(np.uint32)).sum())
            cam.world_view_transform, cam.projection_matrix = view, proj
            cam.FoVx, cam.FoVy, cam.focal_x, cam.focal_y = [float(v) for v in g["cam_fov_focal"][i]]
        return cams

    # hook 2: keep the renderer the job builds
    held = []

    def get_renderer(*a, **k):
        R = orig_get_renderer(*a, **k)
        if t_floor is not None:                    # 0.0 = the reference's semantics to the letter (nothing skipped)
            R.t_floor = float(t_floor)
        held.append(R)
        return R

    settings = g2p.GaussPointCloudSettings(
        renderer_type="python", num_points=num_points, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
        camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
        calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=width,
        max_sh_degree=3, exact_num_points=False, visibility_threshold=0.05, surface_distance_std=None, generate_mesh=False,
        quiet=True, device=str(dev))
    camera_handler.get_cameras, gauss_render.get_renderer = get_cameras, get_renderer
    g2p_get_renderer = getattr(g2p, "get_renderer", None)
    if g2p_get_renderer is not None:
        g2p.get_renderer = get_renderer
    try:
        cloud, _ = g2p.convert_gaussians_to_pc(G, {k: transforms[k] for k in names}, intr, None, settings,
                                               seed=int(g["noise_seed"]))
    finally:
        camera_handler.get_cameras, gauss_render.get_renderer = orig_get_cameras, orig_get_renderer
        if g2p_get_renderer is not None:
            g2p.get_renderer = g2p_get_renderer
    R = held[0]
    out["camera_matrix_bits_differing"] = bits[0]
    out["t_floor"] = float(R.t_floor)
    out["pipeline"] = dict(streams=int(gauss_render.PIPELINE_STREAMS), camera_batch=int(gauss_render.CAMERA_BATCH),
                           rerendered=int(getattr(R, "rerendered", 0)), host_driven=int(getattr(R, "host_driven", 0)))

    # ---- the renderer's final state -------------------------------------------------------------------------------------
    c = R.gaussian_max_contribution.cpu().numpy()
    cs = int(g["contrib_stride"])
    dc = np.abs(c[::cs] - g["contrib_final"])
    out["contrib_max"], out["contrib_frac_gt_1e-4"], out["contrib_compared"] = float(dc.max()), float((dc > 1e-4).mean()), int(dc.shape[0])
    vis, ref_vis = R.get_visible_gaussians().cpu().numpy(), _bits(g["visible_bits"], n)
    flips = np.nonzero(vis != ref_vis)[0]
    out["mask_flips"], out["visible"] = int(flips.size), int(ref_vis.sum())
    out["mask_flip_margins"] = [float(x) for x in np.abs(c[flips] - 0.05)[:16]]
    out["near_threshold_1e-5"] = int((np.abs(g["contrib_final"] - 0.05) < 1e-5).sum())
    # which camera set each running maximum: the slot field of the packed key (contribution << 32 | ~(slot << (12 + seq_bits) | ...))
    key = R.best_key.cpu().numpy().astype(np.uint64)
    low = (~key) & np.uint64(0xFFFFFFFF)
    slot = (low >> np.uint64(12 + int(R.seq_bits))).astype(np.int64)
    seen = c > max(float(R.t_floor), 1e-12)
    ref_w = g["winner_cam"].astype(np.int64)
    wm = seen & (ref_w != 255) & (slot - 1 != ref_w)
    out["winner_camera_mismatch"], out["winner_camera_compared"] = int(wm.sum()), int((seen & (ref_w != 255)).sum())
    cols = (R.get_gaussian_colours().cpu().numpy() / 255.0)[::16]
    dcol = np.abs(cols - g["colours_s16"] / 255.0)[seen[::16]]
    out["colour_max"], out["colour_off_gaussians"] = float(dcol.max()), int((dcol.max(axis=1) > 1e-4).sum())
    out["colour_compared_gaussians"] = int(seen[::16].sum())
    # colours of Gaussians whose winner camera differs are another camera's pixel: reported apart
    ok16 = ~wm[::16][seen[::16]]
    out["colour_max_same_winner"] = float(dcol[ok16].max()) if ok16.any() else 0.0

    # ---- the kept set and the quotas (the job filtered G in place) ------------------------------------------------------
    ref_culled, ref_ppg = _bits(g["culled_bits"], n), g["ppg_u16"].astype(np.int64)
    out["culled_equal"] = bool(np.array_equal(vis, ref_culled))
    kept_n = int(G.xyz.shape[0])
    out["kept"], out["kept_ref"] = kept_n, int(_bits(g["keep_bits"], int(ref_culled.sum())).sum())
    out["keep_equal"] = out["culled_equal"] and kept_n == out["kept_ref"] and bool(_bits(g["keep_bits"], int(ref_culled.sum())).all())
    if out["keep_equal"]:
        contrib = R.get_total_gaussian_contributions()[torch.from_numpy(vis).to(dev)]
        mags = G.get_gaussian_magnitudes(contributions=contrib)
        ppg = ops.distribute_points(mags, num_points)[1].cpu().numpy().astype(np.int64)
        rc = g["kept_contrib"]
        fl, ex, mx, ratio = explain_quota_flips(ppg, ref_ppg, mags.cpu().numpy().astype(np.float64), rc, contrib.cpu().numpy(), num_points)
        out.update(ppg_mismatch_end_to_end=fl, ppg_flips_explained=ex, ppg_max_abs_diff_end_to_end=mx,
                   ppg_flip_margin_over_bound_max=ratio, ppg_compared=int(ppg.shape[0]),
                   kept_contrib_max=float(np.abs(contrib.cpu().numpy() - rc).max()))
        mags2 = G.get_gaussian_magnitudes(contributions=torch.from_numpy(rc).to(dev))
        ppg2 = ops.distribute_points(mags2, num_points)[1].cpu().numpy().astype(np.int64)
        out["ppg_mismatch_given_ref_contrib"] = int((ppg2 != ref_ppg).sum())
        fl2, ex2, mx2, _ = explain_quota_flips(ppg2, ref_ppg, mags2.cpu().numpy().astype(np.float64), rc, rc, num_points)
        out["ppg_flips_explained_given_ref_contrib"], out["ppg_max_abs_diff_given_ref_contrib"] = ex2, mx2
    # ---- the cloud the job returned -----------------------------------------------------------------------------------------
    if sampler:
        pts, rgb = cloud.points.cpu().numpy(), cloud.colours.cpu().numpy()
        out["sample_points"], out["sample_points_ref"] = int(pts.shape[0]), int(g["m"])
        rs = int(g["row_stride"])
        out.update(match_rows(pts, rgb, g["points_s256"], g["colours_s256"], np.arange(0, int(g["m"]), rs), win=64))
    out["reference_cpu_seconds_per_camera_mean"] = float(np.mean(g["seconds_per_camera"]))
    out["reference_cpu_threads"] = int(g["threads"])
    out["check_seconds"] = time.perf_counter() - t_start
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "cuda:0", *(sys.argv[2:3]))))
