"""--generate_mesh path up to (not including) Open3D: the SECOND point cloud, sampled from the predicted-surface Gaussians
only (gauss_to_pc.py:506-509,563-592).  Checked against the CPU oracle of the sampler fed with the surface subset that the
renderer state implies, with the same keyed noise."""
import numpy as np
import torch

import ref_gauss as RG
from np_philox import keyed_normals
from g2pc.synth import make_scene, make_cameras


def check_surface_cloud(device="cpu", n=1500, ncam=3, num_points=30000, seed=31):
    import camera_handler
    import gauss_render
    from gauss_handler import Gaussians
    from gauss_to_pc import GaussPointCloudSettings, convert_gaussians_to_pc
    dev = torch.device(device)
    sc = make_scene(n, 57, scale_lo=0.01, scale_hi=0.06)
    tr, intr = make_cameras(ncam, width=200, height=120, focal=170.0)
    s = GaussPointCloudSettings(
        renderer_type="cuda", num_points=num_points, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
        camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
        calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=None,
        max_sh_degree=3, exact_num_points=False, visibility_threshold=0.05, surface_distance_std=None, generate_mesh=True,
        quiet=True, device=str(dev))
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    total, surface = convert_gaussians_to_pc(G, tr, intr, None, s, seed=seed)
    assert surface is not None and surface.points.shape[0] > 0

    # the same renderer state, rebuilt independently, tells which Gaussians the surface cloud may use
    G2 = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    G2.calculate_normals()
    R = gauss_render.get_renderer("cuda", G2.xyz, G2.opacities.unsqueeze(1), G2.colours, G2.covariances,
                                  visible_gaussian_threshold=0.05, surface_distance_std=None, calculate_surface_distance=True)
    for name in tr:
        R(camera_handler.get_camera("cuda", torch.tensor(tr[name]), intr[name]), return_image=False)
    colours = R.get_gaussian_colours().cpu()
    visible = R.get_visible_gaussians().cpu()
    surf = R.get_predicted_surface_gaussians(predicted_surface_std=1.0).cpu()
    contrib = R.get_total_gaussian_contributions().cpu()
    sel = visible & surf                                     # culled to the visible set first, then to the surface set
    assert 0 < int(sel.sum()) < int(visible.sum())
    cov, keep = RG.validate_covariances(RG.covariances(sc.scales, sc.rots)[sel])
    assert bool(keep.all())
    mesh_points = min(num_points // 2, int(sel.sum()) * 25)
    ref = RG.generate_pointcloud(sc.xyz[sel], cov, colours[sel].float(), RG.normals(sc.scales, sc.rots)[sel], contrib[sel],
                                 mesh_points, std=2.0, exact=False, attempts=5,
                                 eps_fn=lambda g, a, k: keyed_normals(seed, g[:, None], a, np.arange(k)[None, :]))
    got = surface.points.cpu().numpy()
    want = ref["points"].numpy()
    assert abs(got.shape[0] - want.shape[0]) <= 4, (got.shape, want.shape)
    if got.shape == want.shape:
        assert float(np.abs(got - want).max()) < 1e-4
        assert float(np.abs(surface.colours.cpu().numpy() - ref["colours"].numpy()).max()) < 1e-2
    return dict(surface_points=int(got.shape[0]), reference_points=int(want.shape[0]), surface_gaussians=int(sel.sum()),
                total_points=int(total.points.shape[0]))
