"""-m gpu: BASELINE configs[3] sizes on ONE GPU (5 M Gaussians, 50 M points; 3 of the 200 cameras): shakes out the u32
offset / capacity / workspace paths at N = 5 M before an 8-GPU node ever sees the job."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_config4_sizes_on_one_gpu():
    import gauss_render
    from gauss_handler import Gaussians
    from gauss_to_pc import GaussPointCloudSettings, convert_gaussians_to_pc
    from g2pc.synth import make_scene, make_cameras
    n, num_points, ncam = 5_000_000, 50_000_000, 3
    sc = make_scene(n, 1234 + 4, device=DEV)
    tr, intr = make_cameras(200)
    names = sorted(tr)[:ncam]
    cams = ({k: tr[k] for k in names}, {k: intr[k] for k in names})
    s = GaussPointCloudSettings(
        renderer_type="python", num_points=num_points, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
        camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
        calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=1280,
        max_sh_degree=3, exact_num_points=False, visibility_threshold=0.05, surface_distance_std=None, generate_mesh=False,
        quiet=True, device=DEV)
    counts = []
    for rep in range(2):
        g = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours.clone(), sc.opacities)
        cloud, _ = convert_gaussians_to_pc(g, cams[0], cams[1], None, s, seed=9, keep_render_context=False)
        torch.cuda.synchronize()
        m = cloud.points.shape[0]
        counts.append(m)
        assert abs(m - num_points) < 0.002 * num_points, m
        assert bool(torch.isfinite(cloud.points).all()) and float(cloud.points.abs().max()) < 1.2
        assert float(cloud.colours.min()) >= -1e-3 and float(cloud.colours.max()) <= 255.0 + 1e-3
        assert cloud.normals.shape == cloud.points.shape
        L = [x[0] for x in gauss_render.RENDER_STATS[-ncam:]]
        assert min(L) > 5_000_000 and max(L) < 2 ** 31                      # ~4 instances per Gaussian
    assert counts[0] == counts[1]                                            # same seed, same cloud size
