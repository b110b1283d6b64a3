"""The reference's helper functions (eval_sh, build_covariance_2d, projection_ndc, get_radius, get_rect, mahalanobis,
sample_from_multivariate_normal, create_new_gaussian_points, mark_visible) through the CPU emulator."""
from emu_util import emu  # noqa: F401
import helper_checks as H


def test_projection_helpers(emu, golden_dir):
    H.check_projection_helpers(golden_dir)


def test_eval_sh(emu, golden_dir):
    H.check_eval_sh(golden_dir)


def test_mahalanobis_mvn_and_new_points(emu, golden_dir):
    H.check_mahalanobis_and_mvn(golden_dir)


def test_mark_visible(emu):
    H.check_mark_visible()


def test_validate_covariances_cull_branch(emu, golden_dir):
    print(H.check_validate_covariances_cull_branch(golden_dir))


def test_batched_camera_setup_equals_one_camera_at_a_time():
    """camera_handler.get_cameras (one batched inverse for the whole rig) hands out, bit for bit, the matrices get_camera
    computes camera by camera -- both conventions."""
    import torch
    import camera_handler as ch
    from g2pc.synth import make_cameras
    tr, intr = make_cameras(23, width=640, height=360, focal=550.0)
    for kind in ("python", "cuda"):
        rig = ch.get_cameras(kind, tr, intr, colour_resolution=320)
        assert list(rig) == list(tr)
        for k in tr:
            one = ch.get_camera(kind, torch.tensor(tr[k]), intr[k], colour_resolution=320)
            if kind == "python":
                assert torch.equal(rig[k].world_view_transform, one.world_view_transform)
                assert torch.equal(rig[k].projection_matrix, one.projection_matrix)
                assert torch.equal(rig[k].full_proj_transform, one.full_proj_transform)
                assert torch.equal(rig[k].camera_center, one.camera_center)
                assert (rig[k].image_width, rig[k].image_height, rig[k].FoVx, rig[k].focal_y) == (
                    one.image_width, one.image_height, one.FoVx, one.focal_y)
            else:
                for f in ("viewmatrix", "projmatrix", "campos", "bg"):
                    assert torch.equal(getattr(rig[k], f), getattr(one, f)), f
                assert (rig[k].image_height, rig[k].image_width, rig[k].tanfovx, rig[k].tanfovy) == (
                    one.image_height, one.image_width, one.tanfovx, one.tanfovy)
