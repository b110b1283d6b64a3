"""-m gpu: the reference's helper functions on the MI355X (same assertions as tests/test_emu_helpers.py), through the C ABI."""
import pytest

import helper_checks as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_projection_helpers(golden_dir):
    H.check_projection_helpers(golden_dir, DEV)


def test_eval_sh(golden_dir):
    H.check_eval_sh(golden_dir, DEV)


def test_mahalanobis_mvn_and_new_points(golden_dir):
    H.check_mahalanobis_and_mvn(golden_dir, DEV)


def test_mark_visible():
    H.check_mark_visible(DEV)


def test_validate_covariances_cull_branch(golden_dir):
    print(H.check_validate_covariances_cull_branch(golden_dir, DEV))
