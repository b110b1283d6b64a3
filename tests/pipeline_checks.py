"""End-to-end (configs[0]-shaped) parity: 10k Gaussians, 1 camera at colour_resolution=360, 100k points, python
renderer semantics, keyed noise -- against tests/golden/pipeline_cfg1.npz produced by the untouched reference."""
import os

import numpy as np
import torch

from g2pc.synth import make_scene, make_cameras


def run_pipeline_case(golden_dir, device="cpu"):
    import gauss_render
    import camera_handler
    import gauss_to_pc as g2p
    from gauss_handler import Gaussians
    g = np.load(os.path.join(golden_dir, "pipeline_cfg1.npz"))
    dev = torch.device(device)
    sc = make_scene(int(g["n"]), int(g["seed"]), scale_lo=0.004, scale_hi=0.04)
    transforms, intr = make_cameras(1, width=1280, height=720, focal=1100.0)
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    G.calculate_normals()
    R = gauss_render.get_renderer("python", G.xyz, torch.unsqueeze(torch.clone(G.opacities), 1), G.colours,
                                  G.covariances, visible_gaussian_threshold=0.05)
    for name in transforms:
        cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=360)
        R(cam)
    G.colours = R.get_gaussian_colours()
    G.add_gaussians_to_cull(R.get_visible_gaussians())
    G.apply_min_opacity(0.0)
    G.apply_bounding_box(None, None)
    culled = G.filter_gaussians()
    contrib = R.get_total_gaussian_contributions()[culled]
    keep = G.validate_covariances()
    contrib = contrib[keep]
    pts, cols, nrms = g2p.generate_pointcloud(G, int(g["num_points"]), exact_num_points=False,
                                              mahalanobis_distance_std=2.0, calculate_normals=True,
                                              num_sample_attempts=5, contributions=contrib, device=str(dev),
                                              quiet=True, seed=int(g["noise_seed"]))
    return g, culled, keep, contrib, pts, cols, nrms


def assert_pipeline_matches(g, culled, keep, contrib, pts, cols, nrms):
    assert np.array_equal(culled.cpu().numpy(), g["culled"]), "culling mask differs"          # bit-exact indices
    assert np.array_equal(keep.cpu().numpy(), g["keep"])
    np.testing.assert_allclose(contrib.cpu().numpy(), g["contrib"], atol=1e-4)
    p, ref = pts.cpu().numpy(), g["points"]
    # the allocation depends on contributions that agree to ~1e-5, so a handful of Gaussians may get +-1 point:
    # compare counts first, then the clouds when they line up
    assert abs(p.shape[0] - ref.shape[0]) <= 0.001 * ref.shape[0], (p.shape, ref.shape)
    if p.shape == ref.shape:
        d = np.abs(p - ref).max(axis=1)
        frac = float((d > 1e-4).mean())
        assert frac < 0.02, "fraction of points off by > 1e-4: %g" % frac
        if frac == 0.0:
            np.testing.assert_allclose(cols.cpu().numpy(), g["colours"], atol=255e-4)
    return p.shape[0], ref.shape[0]
