"""-m gpu: the reference-shaped native entry (_C.rasterize_gaussians over g2pc_rasterize_gaussians, ABI 7) on the MI355X against
the stored results of the reference's own rasteriser (tests/golden/render_cu_*.npz) -- the GPU twin of
tests/test_emu_c_entry.py; the binding-side reductions (__init__.py:128-158) are restated in tests/c_entry_checks.py."""
import json

import pytest
import torch

import c_entry_checks as CE
import cu_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["n6000_333x187", "n6000_sh3_320x176", "n20000_mask_320x176", "n60000_sh3_1280x720"])
def test_c_entry_matches_reference_fixture_on_gpu(name):
    assert torch.cuda.is_available()
    reps, st, case = CE.drive_fixture(name, "cuda:0")
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)
