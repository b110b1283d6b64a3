"""Shared parity checks for the native-rasteriser ("cuda") semantics.

run_golden_case / assert in cu_golden: the HIP GaussianRasterizer (through the C ABI) against the golden vectors of the
REFERENCE'S OWN rasteriser compiled for the host (tests/golden/render_cu_*.npz, oracle/build_ref.py).
run_cuda_case: the same against the C restatement oracle/cuda_raster_ref.c at sizes that have no fixture -- the restatement
is itself pinned to those fixtures (tests/test_oracle_cuda.py)."""
import numpy as np
import torch

import ref_cuda
import ref_gauss as RG
from g2pc.synth import make_scene, make_cameras


def run_golden_case(name, device="cpu", keep=False):
    """Drives the drop-in exactly like oracle/make_golden_cu.run_case drives the reference; returns (per-camera reports,
    state report, Case) -- and, with keep, the renderer, the Gaussians and the scene for the rest of the conversion."""
    import camera_handler
    import gauss_render
    import cu_golden
    from gauss_handler import Gaussians
    case = cu_golden.Case(name)
    r = case.recipe
    dev = torch.device(device)
    sc, transforms, intr = case.scene()
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    # the reference's own 3-D covariances (its torch.exp is SLEEF's, within an ulp of -- not equal to -- any other exp): the
    # rasteriser is compared on identical inputs; Gaussians' own covariance build is held to the geometry fixtures
    c6 = torch.from_numpy(case.cov6(G.covariances.reshape(-1, 9)[:, [0, 1, 2, 4, 5, 8]].cpu().numpy()))
    cov = c6[:, [0, 1, 2, 1, 3, 4, 2, 4, 5]].reshape(-1, 3, 3).contiguous().to(dev)
    assert float((G.covariances - cov).abs().max()) < 1e-6 * float(cov.abs().max())
    R = gauss_render.get_renderer("cuda", G.xyz, G.opacities.unsqueeze(1), G.colours, cov,
                                  shs=sc.shs.to(dev) if r["with_sh"] else None, visible_gaussian_threshold=0.05,
                                  surface_distance_std=2.0 if r["surf"] else None, calculate_surface_distance=r["surf"])
    reps = []
    for i, nm in enumerate(transforms):
        mask = torch.from_numpy(case.mask.copy()) if case.has_mask else None
        cam = camera_handler.get_camera("cuda", torch.tensor(transforms[nm]), intr[nm], colour_resolution=None, sh_degree=3,
                                        mask=mask)
        # The camera matrices are host arithmetic (torch.linalg.inv + a 4x4 sgemm: MKL picks its kernel by CPU, and the
        # products differ in the last bit between the authoring container and the GPU box): the rasteriser is handed the
        # matrices the reference was handed; get_camera's own result must agree with them to rounding.
        stored = {k: torch.from_numpy(case.cam(i, k)) for k in ("viewmatrix", "projmatrix", "campos")}
        for k, v in stored.items():
            assert torch.allclose(getattr(cam, k), v, rtol=1e-5, atol=1e-6), k
        assert abs(cam.tanfovx - float(case.cam(i, "tanfovx"))) < 1e-7 and abs(cam.tanfovy - float(case.cam(i, "tanfovy"))) < 1e-7
        cam = cam._replace(tanfovx=float(case.cam(i, "tanfovx")), tanfovy=float(case.cam(i, "tanfovy")), **stored)
        colour, radii, invd, dep = R.forward(cam, return_per_camera=True)
        rad = radii.cpu().numpy()
        rc = R._sync.rect.cpu().numpy().astype(np.int64)
        if max(case.recipe["width"], case.recipe["height"]) > 4096:  # two words: x0 | (x1-1) << 16, y0 | (y1-1) << 16
            rx, ry = rc[0:2 * len(rad):2], rc[1:2 * len(rad):2]
            touched = np.where(rad > 0, ((rx >> 16) - (rx & 0xFFFF) + 1) * ((ry >> 16) - (ry & 0xFFFF) + 1), 0)
        else:                                                        # x0 | (x1-1) << 8 | y0 << 16 | (y1-1) << 24, 0 = no tile
            rc = rc[:len(rad)]
            touched = np.where(rad > 0, (((rc >> 8) & 255) - (rc & 255) + 1) * (((rc >> 24) & 255) - ((rc >> 16) & 255) + 1), 0)
        rec = R._sync.rec.cpu().numpy().reshape(-1, 16)             # (px, py, qa, qb), (qc, opacity, depth, radius), ...
        got = dict(radii=rad, num_rendered=R.last["num_rendered"], tiles_touched=touched.astype(np.uint32),
                   means2D=rec[:, 0:2], depths=rec[:, 6], conic_scaled=rec[:, 2:5], opacity=rec[:, 5], rgb=rec[:, 8:11],
                   out_color=colour.cpu().numpy(), out_depth=dep.cpu().numpy(), out_invdepth=invd.cpu().numpy(),
                   gauss_contributions=R.last["contributions"].cpu().numpy(), gauss_pixels=R.last["pixels"].cpu().numpy(),
                   gauss_surface_distances=R.last["surface_distances"].cpu().numpy())
        reps.append(cu_golden.compare_camera(case, i, got))
    st = dict(max_contribution=R.gaussian_max_contribution.cpu().numpy(),
              total_contribution=R.get_total_gaussian_contributions().cpu().numpy(),
              min_surface_distance=R.gaussian_min_surface_distance.cpu().numpy(),
              colours=R.get_gaussian_colours().cpu().numpy(), visible=R.get_visible_gaussians().cpu().numpy())
    if r["surf"]:
        st["low_surface_distance"] = R.get_gaussians_with_low_surface_distance().cpu().numpy()
        st["predicted_surface"] = R.get_predicted_surface_gaussians(0.5).cpu().numpy()
    if keep:
        return reps, cu_golden.compare_state(case, st), case, R, G, sc
    return reps, cu_golden.compare_state(case, st), case


def run_cuda_case(n, seed, width, height, focal, ncam, device="cpu", with_sh=False, surf=True, scale=(0.004, 0.04),
                  use_mask=False, colour_resolution=None):
    import camera_handler
    import gauss_render
    from gauss_handler import Gaussians
    dev = torch.device(device)
    sc = make_scene(n, seed, with_sh=with_sh, scale_lo=scale[0], scale_hi=scale[1])
    transforms, intr = make_cameras(ncam, width=width, height=height, focal=focal)
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    R = gauss_render.get_renderer("cuda", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                  shs=sc.shs.to(dev) if with_sh else None, visible_gaussian_threshold=0.05,
                                  surface_distance_std=2.0 if surf else None, calculate_surface_distance=surf)
    cov6 = RG.strip_symmetric(RG.covariances(sc.scales, sc.rots)).numpy()
    O = ref_cuda.CudaRasterizerOracle(sc.xyz.numpy(), sc.opacities.numpy(), cov6,
                                      colors_precomp=None if with_sh else sc.colours.numpy(),
                                      shs=sc.shs.numpy() if with_sh else None, sh_degree=3, threshold=0.05,
                                      surface_distance_std=2.0 if surf else None, calculate_surface_distance=surf)
    rep = dict(image=0.0, depth=0.0, invdepth=0.0, contrib=0.0, surf_frac_off=0.0, radii_mismatch=0, pix_mismatch=0,
               num_rendered=[])
    rng = np.random.default_rng(seed)
    for name in transforms:
        mask = None
        if use_mask:
            mask = torch.from_numpy((rng.random((height, width)) > 0.3).astype(np.int32))
        cam = camera_handler.get_camera("cuda", torch.tensor(transforms[name]), intr[name],
                                        colour_resolution=colour_resolution, sh_degree=3, mask=mask)
        colour, radii, invd, dep = R.forward(cam, return_per_camera=True)
        o = O.forward(ref_cuda.camera_settings(transforms[name], intr[name], colour_resolution if not use_mask else None),
                      mask=None if mask is None else mask.numpy().reshape(-1))
        rep["num_rendered"].append((R.last["num_rendered"], o["num_rendered"]))
        di = np.abs(colour.cpu().numpy() - o["colour"])
        rep["image"] = max(rep["image"], float(di.max()))
        rep["image_frac_off"] = max(rep.get("image_frac_off", 0.0), float((di > 2e-4).mean()))
        rep["depth"] = max(rep["depth"], float(np.abs(dep.cpu().numpy() - o["depth"]).max()))
        rep["invdepth"] = max(rep["invdepth"], float(np.abs(invd.cpu().numpy() - o["invdepth"]).max()))
        rep["radii_mismatch"] += int((radii.cpu().numpy() != o["radii"]).sum())
        c = R.last["contributions"].cpu().numpy()
        rep["contrib"] = max(rep["contrib"], float(np.abs(c - o["contrib"]).max()))
        rep["contrib_frac_off"] = max(rep.get("contrib_frac_off", 0.0), float((np.abs(c - o["contrib"]) > 1e-4).mean()))
        same = np.abs(c - o["contrib"]) < 1e-6
        rep["pix_mismatch"] += int(((R.last["pixels"].cpu().numpy() != o["pixels"]) & same & (o["contrib"] > 0)).sum())
        if surf:
            s, so = R.last["surface_distances"].cpu().numpy(), o["surf"]
            fin = (so < 3e38) | (s < 3e38)
            off = np.abs(np.where(fin, s, 0) - np.where(fin, so, 0)) > 1e-4 * np.maximum(1.0, np.abs(np.where(fin, so, 0)))
            rep["surf_frac_off"] = max(rep["surf_frac_off"], float(off.mean()))
    rep["state_max"] = float(np.abs(R.gaussian_max_contribution.cpu().numpy() - O.max_contribution).max())
    rep["state_total"] = float(np.abs(R.get_total_gaussian_contributions().cpu().numpy() - O.total).max())
    dcol = np.abs(R.get_gaussian_colours().cpu().numpy() - O.get_gaussian_colours()) / 255.0
    rep["state_colour"] = float(dcol.max())
    rep["state_colour_frac_off"] = float((dcol > 2e-4).mean())
    rep["visible_flips"] = int((R.get_visible_gaussians().cpu().numpy() != O.get_visible_gaussians()).sum())
    if surf:
        rep["surface_mask_flips"] = int((R.get_gaussians_with_low_surface_distance().cpu().numpy() !=
                                         O.get_surface_gaussians_below_distance_threshold(2.0)).sum())
    return rep


def assert_cuda_matches(rep, n):
    for a, b in rep["num_rendered"]:
        assert abs(a - b) <= max(2, 1e-4 * b), rep["num_rendered"]       # instance counts (tile rect is integer maths)
    assert rep["radii_mismatch"] <= max(1, n // 20000), rep
    # the alpha < 1/255 and T(1-alpha) < 1e-4 cut-offs are threshold decisions on fp32 values: a last-bit difference
    # (v_exp_f32 vs expf) flips a whole term of up to ~alpha*T*c at an isolated pixel; the bulk agrees to ~1e-6
    assert rep["image_frac_off"] < 2e-3 and rep["image"] < 2e-2 and rep["depth"] < 0.1 and rep["invdepth"] < 2e-2, rep
    # ... and the same flips move T for everything behind the flipped term at that pixel
    assert rep["contrib_frac_off"] < 1e-3 and rep["contrib"] < 2e-2 and rep["state_max"] < 2e-2, rep
    assert rep["state_colour_frac_off"] < 1e-3 and rep["state_colour"] < 5e-2, rep
    assert rep["state_total"] < 5e-2, rep
    assert rep["pix_mismatch"] <= max(2, n // 2000), rep
    assert rep["visible_flips"] <= 1, rep
    if "surface_mask_flips" in rep:
        assert rep["surf_frac_off"] < 2e-3, rep
        assert rep["surface_mask_flips"] <= max(2, n // 2000), rep


def run_configs4_end_to_end(device, golden_dir):
    """BASELINE configs[4] END TO END at its own size against tests/golden/pipeline_cu_cfg4_1m.npz (oracle/make_golden_cu.py
    --e2e: the reference's own rasteriser for cameras 0 and 17, then the reference's own conversion tail -- surface cull,
    unrendered cull, filter, validate, generate_pointcloud(exact_num_points=True, 100 attempts) with keyed noise).
    Returns the gates; tests/test_gpu_cuda_semantics.py asserts them."""
    import os
    import sys
    import gauss_to_pc as g2p
    from gauss_handler import Gaussians
    from g2pc import ops
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from parity_cfg2 import match_rows
    f = np.load(os.path.join(golden_dir, "pipeline_cu_cfg4_1m.npz"))
    n = int(f["n"])
    bits = lambda k, m=n: np.unpackbits(f[k])[:m].astype(bool)
    reps, st, case, R, G, sc = run_golden_case(str(f["case"]), device, keep=True)
    dev = torch.device(device)
    out = dict(gaussians=n, kept_ref=int(f["kept"]), points_ref=int(f["m"]))
    # ---- the cull chain from OUR render (gauss_to_pc.py:478-546 order) ----
    G.colours = R.get_gaussian_colours()
    low, vis = R.get_gaussians_with_low_surface_distance(), R.get_visible_gaussians()
    out["low_surface_flips"] = int((low.cpu().numpy() != bits("low_surface_bits")).sum())
    out["visible_flips"] = int((vis.cpu().numpy() != bits("visible_bits")).sum())
    G.add_gaussians_to_cull(low)
    G.add_gaussians_to_cull(vis)
    G.apply_min_opacity(0.0)
    G.apply_bounding_box(None, None)
    G.cull_large_gaussians(0.0)
    culled = G.filter_gaussians()
    out["culled_equal"] = bool(np.array_equal(culled.cpu().numpy(), bits("culled_bits")))
    contrib = R.get_total_gaussian_contributions()[culled]
    keep = G.validate_covariances()
    kb = bits("keep_bits", int(keep.numel()))
    out["keep_equal"] = bool(np.array_equal(keep.cpu().numpy().astype(bool), kb)) if keep.numel() == kb.shape[0] else False
    ref_ppg = f["ppg"].astype(np.int64)
    if out["culled_equal"] and out["keep_equal"]:
        contrib = contrib[keep]
        out["contrib_max"] = float((contrib.cpu() - torch.from_numpy(f["kept_contrib"])).abs().max())
        mags = G.get_gaussian_magnitudes(contributions=contrib)
        ppg = ops.distribute_points(mags, int(f["num_points"]))[1].cpu().numpy().astype(np.int64)
        out["ppg_mismatch_end_to_end"] = int((ppg != ref_ppg).sum())
        out["ppg_max_abs_diff_end_to_end"] = int(np.abs(ppg - ref_ppg).max())
    del R
    # ---- quotas and the exact-points sampler on the REFERENCE's kept set (isolates distribute_points and the 100-attempt sampler) ----
    rc = torch.from_numpy(bits("culled_bits"))
    G2 = Gaussians(sc.xyz[rc].to(dev), sc.scales[rc].to(dev), sc.rots[rc].to(dev), torch.from_numpy(f["kept_colours"]).to(dev),
                   sc.opacities[rc].to(dev))
    keep2 = G2.validate_covariances()
    out["keep_all_of_ref_kept"] = bool(keep2.all()) and int(keep2.numel()) == int(bits("keep_bits", int(keep2.numel())).sum())
    kc = torch.from_numpy(f["kept_contrib"]).to(dev)
    mags2 = G2.get_gaussian_magnitudes(contributions=kc)
    ppg2 = ops.distribute_points(mags2, int(f["num_points"]))[1].cpu().numpy().astype(np.int64)
    out["ppg_mismatch_given_ref_contrib"] = int((ppg2 != ref_ppg).sum())
    out["ppg_max_abs_diff_given_ref_contrib"] = int(np.abs(ppg2 - ref_ppg).max())
    pts, cols, _ = g2p.generate_pointcloud(G2, int(f["num_points"]), exact_num_points=True, mahalanobis_distance_std=2.0,
                                           calculate_normals=False, num_sample_attempts=100, contributions=kc, device=str(dev),
                                           quiet=True, seed=int(f["noise_seed"]))
    out["sample_points"] = int(pts.shape[0])
    rs = int(f["row_stride"])
    out.update(match_rows(pts.cpu().numpy(), cols.cpu().numpy(), f["points_s256"], f["colours_s256"], np.arange(0, int(f["m"]), rs)))
    return reps, st, case, out
