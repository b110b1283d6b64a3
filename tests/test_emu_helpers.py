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
