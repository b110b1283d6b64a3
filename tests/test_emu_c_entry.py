"""The reference-shaped native entry (_C.rasterize_gaussians / mark_visible over g2pc_rasterize_gaussians, ABI 7) on the CPU
emulator build of the product sources.

 * against the golden vectors of the reference's own rasteriser (every integer equal, floats < 1e-5), and
 * (-m reference, container only) the reference's UNMODIFIED gaussian_pointcloud_rasterization/__init__.py imported twice --
   over oracle/_ref/synced_nofma/_C.so (its own CUDA sources compiled for the host) and over this package's _C.py -- driven
   through the same cameras: running state, images, radii and getter masks compared."""
import json

import numpy as np
import pytest
import torch

import c_entry_checks as CE
import cu_golden
from emu_util import emu  # noqa: F401


@pytest.mark.parametrize("name", ["n6000_333x187", "n6000_sh3_320x176", "n20000_mask_320x176"])
def test_c_entry_matches_reference_fixture(emu, name):
    reps, st, case = CE.drive_fixture(name, "cpu")
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)


def test_c_entry_argument_contract(emu):
    _C = CE.load_c_module()
    e = torch.Tensor([])
    eye = torch.eye(4)
    base = dict(background=torch.ones(3), means3D=torch.zeros((4, 3)), colors=torch.zeros((4, 3)), opacity=torch.ones((4, 1)),
                scales=e, rotations=e, scale_modifier=1.0, cov3D_precomp=torch.zeros((4, 6)), viewmatrix=eye, projmatrix=eye,
                tan_fovx=0.5, tan_fovy=0.5, image_height=32, image_width=48, sh=e, degree=0, campos=torch.zeros(3),
                mask=torch.ones(32 * 48, dtype=torch.int32), prefiltered=False, antialiasing=False,
                calculate_surface_distance=False, debug=False)
    order = list(base)
    assert len(order) == 22
    call = lambda **kw: _C.rasterize_gaussians(*[dict(base, **kw)[k] for k in order])
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):          # rasterize_points.cu:61-63
        call(means3D=torch.zeros((4, 2)))
    with pytest.raises(RuntimeError):                                                 # neither colours nor SHs
        call(colors=e)
    with pytest.raises(RuntimeError):                                                 # covariance AND scale / rotation pair
        call(scales=torch.ones((4, 3)), rotations=torch.ones((4, 4)))
    # P == 0 (rasterize_points.cu:101): zero images, empty per-Gaussian results, nothing rendered
    out = call(means3D=torch.zeros((0, 3)), colors=torch.zeros((0, 3)), opacity=torch.zeros((0, 1)), cov3D_precomp=torch.zeros((0, 6)))
    assert out[0] == 0 and float(out[1].abs().max()) == 0.0 and out[3].numel() == 0 and out[8].numel() == 0
    # Gaussians behind the camera: radii 0, nothing rendered, the background everywhere
    out = call(means3D=torch.tensor([[0.0, 0.0, -1.0]] * 4))
    assert out[0] == 0 and int(out[3].abs().sum()) == 0 and float(out[8].max()) == 0.0
    assert float(out[9].min()) == pytest.approx(float(np.finfo(np.float32).max)) and torch.all(out[1] == 1.0)
    vis = _C.mark_visible(torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 0.1], [0.0, 0.0, -1.0]]), eye, eye)
    assert vis.dtype == torch.bool and vis.tolist() == [True, False, False]
    assert _C.mark_visible(torch.zeros((0, 3)), eye, eye).numel() == 0


@pytest.mark.reference
@pytest.mark.parametrize("mode", ["cov", "scales", "antialiasing"])
def test_unmodified_reference_binding_runs_on_the_hip_library(emu, mode):
    """the reference's own __init__.py, untouched, over (a) the reference's own rasteriser and (b)
    libg2pc through _C.py."""
    import build_ref
    import os
    if not build_ref.available():
        pytest.skip("reference sources absent (GPU box)")
    ref_dir = build_ref.build("synced", False)
    ours_dir = os.path.join(CE.PKG, CE.GPR_NAME)
    case = cu_golden.Case("n6000_333x187")
    kw = dict(with_scales=(mode != "cov"), antialiasing=(mode == "antialiasing"))
    a = CE.run_reference_binding(CE.load_reference_binding_over(ref_dir), case, **kw)
    b = CE.run_reference_binding(CE.load_reference_binding_over(ours_dir), case, **kw)
    assert np.array_equal(a["radii"], b["radii"])
    assert np.array_equal(a["visible"], b["visible"]) and np.array_equal(a["low_surface"], b["low_surface"])
    rep = dict(max=float(np.abs(a["max"] - b["max"]).max()), total=float(np.abs(a["total"] - b["total"]).max()),
               colours=float(np.abs(a["colours"] - b["colours"]).max() / 255.0), image=float(np.abs(a["image"] - b["image"]).max()),
               depth=float(np.abs(a["depth"] - b["depth"]).max()), invdepth=float(np.abs(a["invdepth"] - b["invdepth"]).max()),
               reached=int(((a["min_surf"] < 3e38) != (b["min_surf"] < 3e38)).sum()))
    fin = (a["min_surf"] < 3e38) & (b["min_surf"] < 3e38)
    rep["min_surf_rel"] = float((np.abs(a["min_surf"] - b["min_surf"])[fin] / np.maximum(1.0, np.abs(a["min_surf"][fin]))).max())
    print(mode, json.dumps(rep))
    assert rep["max"] < 1e-5 and rep["total"] < 2e-5 and rep["colours"] < 1e-5 and rep["image"] < 1e-5, rep
    assert rep["depth"] < 5e-5 and rep["invdepth"] < 1e-5 and rep["reached"] == 0 and rep["min_surf_rel"] < 1e-4, rep
    if mode == "antialiasing":           # ... and the flag does something (opacities scaled by <= 1, forward.cu:224-225,264)
        c = CE.run_reference_binding(CE.load_reference_binding_over(ours_dir), case, with_scales=True, antialiasing=False)
        assert float(np.abs(c["max"] - b["max"]).max()) > 1e-2
