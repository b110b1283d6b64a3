"""Native-rasteriser ("cuda") semantics through the fiber emulator vs oracle/cuda_raster_ref.c."""
import pytest

from emu_util import emu  # noqa: F401
from cuda_checks import run_cuda_case, assert_cuda_matches


def test_cuda_semantics_precomputed_colours_with_surface_distance(emu):
    rep = run_cuda_case(1500, 5, 200, 120, 170.0, 2, surf=True, scale=(0.01, 0.06))
    print(rep)
    assert_cuda_matches(rep, 1500)


def test_cuda_semantics_sh_and_mask(emu):
    rep = run_cuda_case(800, 6, 168, 100, 150.0, 1, with_sh=True, surf=False, scale=(0.01, 0.06), use_mask=True)
    print(rep)
    assert_cuda_matches(rep, 800)
