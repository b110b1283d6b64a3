import pytest

from emu_util import emu  # noqa: F401
from pipeline_checks import run_pipeline_case, assert_pipeline_matches


def test_pipeline_config1_matches_reference(emu, golden_dir):
    out = run_pipeline_case(golden_dir)
    print(assert_pipeline_matches(*out))
