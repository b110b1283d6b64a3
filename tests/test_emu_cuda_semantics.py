"""Native-rasteriser ("cuda") semantics through the fiber emulator: against the golden vectors of the reference's own
rasteriser (tests/golden/render_cu_*.npz) and, at other sizes, against the restatement pinned to them."""
import json

import pytest

import cu_golden
from emu_util import emu  # noqa: F401
from cuda_checks import run_cuda_case, assert_cuda_matches, run_golden_case


@pytest.mark.parametrize("name", ["n6000_333x187", "n6000_sh3_320x176", "n20000_mask_320x176", "n6000_nosurf_333x187",
                                  "n40000_wide_4128x48"])
def test_cuda_semantics_vs_reference_golden(emu, name):
    """radii / num_rendered / tiles_touched / arg-max pixels bit for bit, floats to 1e-4 (tests/cu_golden.py)."""
    reps, st, case = run_golden_case(name)
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)


def test_cuda_semantics_precomputed_colours_with_surface_distance(emu):
    rep = run_cuda_case(1500, 5, 200, 120, 170.0, 2, surf=True, scale=(0.01, 0.06))
    print(rep)
    assert_cuda_matches(rep, 1500)


def test_cuda_semantics_sh_and_mask(emu):
    rep = run_cuda_case(800, 6, 168, 100, 150.0, 1, with_sh=True, surf=False, scale=(0.01, 0.06), use_mask=True)
    print(rep)
    assert_cuda_matches(rep, 800)


def test_mark_visible_is_the_near_plane_test(emu):
    """_C.mark_visible (rasterize_points.cu:147-166 -> in_frustum, auxiliary.h:151-176): z_view > 0.2 and nothing else."""
    import numpy as np
    import torch
    import camera_handler
    from gaussian_pointcloud_rasterization import GaussianRasterizer
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(3000, 3, scale_lo=0.01, scale_hi=0.05)
    tr, intr = make_cameras(1, width=160, height=90, focal=140.0, radius=0.6)      # camera inside the cloud
    name = next(iter(tr))
    rs = camera_handler.get_camera("cuda", torch.tensor(tr[name]), intr[name])
    R = GaussianRasterizer(sc.xyz, torch.zeros_like(sc.xyz), sc.opacities.unsqueeze(1), colors_precomp=sc.colours,
                           scales=torch.exp(sc.scales), rotations=sc.rots)
    vis = R.markVisible(sc.xyz, rs).numpy()
    V = rs.viewmatrix.numpy().astype(np.float32)
    x = sc.xyz.numpy()
    z = (V[0, 2] * x[:, 0] + V[1, 2] * x[:, 1] + V[2, 2] * x[:, 2] + V[3, 2]).astype(np.float32)
    assert vis.dtype == np.bool_ and np.array_equal(vis, z > np.float32(0.2))
    assert 0 < vis.sum() < vis.size                                               # both sides of the plane are present


def test_generate_mesh_surface_point_cloud(emu):
    from mesh_surface_checks import check_surface_cloud
    print(check_surface_cloud())


@pytest.mark.parametrize("fused", [True, False])
def test_cameras_without_read_back_equal_one_at_a_time(emu, monkeypatch, fused):
    """The pipelined native-semantics path (instance count kept on the device, launches sized for a capacity) leaves the same
    running state as one camera at a time -- including cameras that outgrow the capacity learned from the first one, which are
    skipped on the device and rendered again through the two-call path.  fused: ONE call per camera, g2pc_raster_camera_cu (the
    depth bucket sort emits the instances itself); else g2pc_raster_front_cu + g2pc_raster_back_cu_dev (radix sort, scan,
    k_duplicate, k_resolve_count) as until round 4."""
    import torch
    import camera_handler
    import gaussian_pointcloud_rasterization as gpr
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(2500, 11, scale_lo=0.01, scale_hi=0.06)
    tr, intr = make_cameras(6, width=176, height=100, focal=150.0)
    names = sorted(tr)

    def run(pipelined, shrink):
        monkeypatch.setattr(gpr, "PIPELINE_IN_EMULATOR", pipelined)
        monkeypatch.setattr(gpr, "FUSED_CAMERA_CALL", fused)
        monkeypatch.setattr(gpr, "MIN_CAPACITY", 16)
        monkeypatch.setattr(gpr, "PIPELINE_STREAMS", 2)        # retire early: the capacity grows while cameras are still coming
        R = gpr.GaussianRasterizer(sc.xyz, torch.zeros_like(sc.xyz), sc.opacities.unsqueeze(1), colors_precomp=sc.colours,
                                   scales=torch.exp(sc.scales), rotations=sc.rots, visible_gaussian_threshold=0.05,
                                   surface_distance_std=2.0, calculate_surface_distance=True)
        for i, k in enumerate(names):
            R(camera_handler.get_camera("cuda", torch.tensor(tr[k]), intr[k]), return_image=False)
            if i == 0 and pipelined:
                R._capacity = int(R._capacity * shrink)        # what the first camera taught is too small for the next ones
        R.flush()
        return R, (R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(), R.gaussian_colours.clone(),
                   R.gaussian_min_surface_distance.clone())

    _, ref = run(False, 1.0)
    R2, b = run(True, 0.7)                 # capacity = 87 % of the first camera's count: the next cameras overflow it
    assert R2._capacity is not None and 1 <= R2.rerendered < 4       # some cameras outgrew the capacity, later ones fitted the grown one
    for i, (x, z) in enumerate(zip(ref, b)):
        if i == 1:      # the running SUM of the per-camera maxima: a re-rendered camera is added out of order (fp32 rounding)
            assert torch.allclose(x, z, rtol=1e-6, atol=1e-7)
        else:
            assert torch.equal(x, z)


def test_pipelined_path_on_an_image_wider_than_4096_pixels(emu, monkeypatch):
    """258 x 3 tiles (4 128 x 48 pixels): 16-bit tile coordinates, two rect words per Gaussian, through the device-resident
    binning of the pipelined path as well as through the two-call path -- the same running state."""
    import torch
    import camera_handler
    import gaussian_pointcloud_rasterization as gpr
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(20000, 9, scale_lo=0.004, scale_hi=0.03)
    tr, intr = make_cameras(3, width=4128, height=48, focal=3600.0)

    def run(pipelined):
        monkeypatch.setattr(gpr, "PIPELINE_IN_EMULATOR", pipelined)
        monkeypatch.setattr(gpr, "PIPELINE_STREAMS", 2)
        R = gpr.GaussianRasterizer(sc.xyz, torch.zeros_like(sc.xyz), sc.opacities.unsqueeze(1), colors_precomp=sc.colours,
                                   scales=torch.exp(sc.scales), rotations=sc.rots, visible_gaussian_threshold=0.05,
                                   surface_distance_std=2.0, calculate_surface_distance=True)
        for k in sorted(tr):
            R(camera_handler.get_camera("cuda", torch.tensor(tr[k]), intr[k]), return_image=False)
        R.flush()
        return (R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(), R.gaussian_colours.clone(),
                R.gaussian_min_surface_distance.clone())

    a, b = run(False), run(True)
    assert float(a[0].max()) > 0.05                      # something was really rendered
    for x, z in zip(a, b):
        assert torch.equal(x, z)


def test_fused_camera_call_with_a_depth_pile_up_and_sh(emu, monkeypatch):
    """g2pc_raster_camera_cu on a sheet of Gaussians seen head on (one depth bucket beyond its room: the camera is skipped and
    rendered again through the two-call path) and with SH colours + a mask: the running state of one camera at a time."""
    import torch
    import camera_handler
    import gaussian_pointcloud_rasterization as gpr
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(2600, 13, with_sh=True, scale_lo=0.01, scale_hi=0.05)
    xyz = sc.xyz.clone()
    xyz[:1500, 2] = 0.0
    tr, intr = make_cameras(3, width=160, height=96, focal=140.0)
    eye = torch.eye(4)
    eye[2, 3] = 3.5
    cams = [torch.tensor(tr[sorted(tr)[0]]), eye.clone(), torch.tensor(tr[sorted(tr)[2]]), eye.clone()]
    cams[3][0, 3] = 0.15
    mask = torch.ones((96, 160), dtype=torch.int32)
    mask[:, :48] = 0                                                    # whole 16-pixel tile columns: defined in the reference

    def run(pipelined):
        monkeypatch.setattr(gpr, "PIPELINE_IN_EMULATOR", pipelined)
        monkeypatch.setattr(gpr, "PIPELINE_STREAMS", 2)
        R = gpr.GaussianRasterizer(xyz, torch.zeros_like(xyz), sc.opacities.unsqueeze(1), shs=sc.shs,
                                   scales=torch.exp(sc.scales), rotations=sc.rots, visible_gaussian_threshold=0.05,
                                   surface_distance_std=2.0, calculate_surface_distance=True)
        for c2w in cams:
            R(camera_handler.get_camera("cuda", c2w, intr[sorted(intr)[0]], sh_degree=3, mask=mask), return_image=False)
        R.flush()
        return R, (R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(), R.gaussian_colours.clone(),
                   R.gaussian_min_surface_distance.clone())

    _, a = run(False)
    R2, b = run(True)
    assert R2.rerendered >= 1                                           # the pile-up really happened and was handled
    assert float(a[0].max()) > 0.05
    for i, (x, z) in enumerate(zip(a, b)):
        if i == 1:
            assert torch.allclose(x, z, rtol=1e-6, atol=1e-7)
        else:
            assert torch.equal(x, z)
