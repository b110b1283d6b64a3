"""The native-rasteriser ("cuda") oracle, pinned.

1. tests/golden/render_cu_*.npz hold outputs of the reference's OWN rasteriser (its .cu files compiled for the host,
   oracle/build_ref.py + oracle/make_golden_cu.py).  Where /root/reference exists (authoring container) the smallest
   fixture is regenerated from the sources and must come out bit for bit -- the fixtures are what the sources compute.
2. oracle/cuda_raster_ref.c, the plain-C restatement used at sizes that have no fixture (tests/cuda_checks.run_cuda_case,
   bench.py's cpu_baseline), is held to every fixture with the bars of tests/cu_golden.py.
3. Sanity of the restatement against the (reference-pinned) python-renderer oracle, as before."""
import json
import os

import numpy as np
import pytest
import torch

import cu_golden
import ref_cuda
import ref_gauss as RG
import ref_render as RR
from g2pc.synth import make_scene, make_cameras


def _restatement_on(case):
    sc, transforms, intr = case.scene()
    r = case.recipe
    cov6 = case.z["state_cov6"]              # the covariances and camera matrices the reference was handed (host
    assert np.allclose(cov6, RG.strip_symmetric(RG.covariances(sc.scales, sc.rots)).numpy(), rtol=1e-5, atol=1e-9)
    O = ref_cuda.CudaRasterizerOracle(sc.xyz.numpy(), sc.opacities.numpy(), cov6,
                                      colors_precomp=None if r["with_sh"] else sc.colours.numpy(),
                                      shs=sc.shs.numpy() if r["with_sh"] else None, sh_degree=3, threshold=0.05,
                                      surface_distance_std=2.0 if r["surf"] else None, calculate_surface_distance=r["surf"])
    reps = []
    for i, nm in enumerate(transforms):
        cam = ref_cuda.camera_settings(transforms[nm], intr[nm])
        for k in ("viewmatrix", "projmatrix", "campos"):                      # arithmetic: last bits differ between hosts)
            assert np.allclose(cam[k], case.cam(i, k), rtol=1e-5, atol=1e-6), k
            cam[k] = case.cam(i, k)
        cam["tanfovx"], cam["tanfovy"] = float(case.cam(i, "tanfovx")), float(case.cam(i, "tanfovy"))
        o = O.forward(cam, mask=case.mask.reshape(-1) if case.has_mask else None)
        reps.append(cu_golden.compare_camera(case, i, dict(
            radii=o["radii"], num_rendered=o["num_rendered"], out_color=o["colour"], out_depth=o["depth"],
            out_invdepth=o["invdepth"], gauss_contributions=o["contrib"], gauss_pixels=o["pixels"],
            gauss_surface_distances=o["surf"])))
    st = dict(max_contribution=O.max_contribution, total_contribution=O.total, min_surface_distance=O.min_surface,
              colours=O.get_gaussian_colours(), visible=O.get_visible_gaussians())
    if r["surf"]:
        st["low_surface_distance"] = O.get_surface_gaussians_below_distance_threshold(2.0)
        st["predicted_surface"] = O.get_surface_gaussians_below_distance_threshold(0.5)
    return reps, cu_golden.compare_state(case, st)


@pytest.mark.parametrize("name", cu_golden.CASES)
def test_restatement_is_pinned_to_the_reference(name):
    """oracle/cuda_raster_ref.c against every fixture: integers equal, floats within 1e-5 (both are compiled without
    floating-point contraction and evaluate the same expressions in the same order)."""
    case = cu_golden.Case(name)
    reps, st = _restatement_on(case)
    for rep in reps:
        print(json.dumps(rep))
        cu_golden.assert_camera(rep, case)
    print(json.dumps(st))
    cu_golden.assert_state(st, case)


@pytest.mark.reference
def test_fixture_is_what_the_reference_sources_compute():
    import build_ref
    if not build_ref.available():
        pytest.skip("reference sources absent (GPU box): the fixtures stand for them")
    import make_golden_cu
    import ref_shim
    name = "n6000_nosurf_333x187"
    case = cu_golden.Case(name)
    assert str(case.z["reference_digest"]) == build_ref.source_digest(), "reference changed: regenerate tests/golden/render_cu_*"
    cams, state, _ = make_golden_cu.run_case(ref_shim.load_reference(), name, "synced", False)
    for i, d in enumerate(cams):
        assert d["num_rendered"] == int(case.cam(i, "num_rendered"))
        for k in ("radii", "tiles_touched", "ranges", "gauss_contributions", "gauss_pixels", "means2D", "conic_opacity"):
            assert np.array_equal(d[k].reshape(-1), case.cam(i, k).reshape(-1)), k
        assert np.array_equal(d["out_color"].reshape(3, -1), case.cam(i, "out_color"))
    assert np.array_equal(state["colours"], case.z["state_colours"])


def test_fixture_spread_is_recorded():
    """Every fixture carries how far the reference lands from itself under another legal compilation (no FMA contraction)
    and under the thread_rank schedule without the two inserted barriers (oracle/build_ref.py): the basis of the bars."""
    for name in cu_golden.CASES:
        sp = cu_golden.Case(name).spread
        assert set(sp) == {"verbatim_nofma", "synced_fma", "verbatim_fma"}
        assert sp["verbatim_nofma"]["image_max"] == 0.0 and sp["verbatim_nofma"]["radii_mismatch"] == 0     # races touch no pixel
        assert sp["verbatim_nofma"]["contrib_higher"] == 0 and sp["verbatim_nofma"]["contrib_lower"] > 0     # lower bounds only
        assert sp["synced_fma"]["contrib_frac_gt_1e4"] == 0.0 and sp["synced_fma"]["visible_flips"] == 0


def test_cuda_oracle_agrees_with_python_renderer_oracle_up_to_semantics():
    sc = make_scene(3000, 9, scale_lo=0.004, scale_hi=0.04)
    cov = RG.covariances(sc.scales, sc.rots)
    tr, intr = make_cameras(1, width=320, height=180, focal=275.0)
    name = next(iter(tr))
    O = ref_cuda.CudaRasterizerOracle(sc.xyz.numpy(), sc.opacities.numpy(), RG.strip_symmetric(cov).numpy(),
                                      colors_precomp=sc.colours.numpy(), calculate_surface_distance=True)
    out = O.forward(ref_cuda.camera_settings(tr[name], intr[name]))
    P = RR.PythonRendererOracle(sc.xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05)
    img = P(RR.get_camera(torch.tensor(tr[name]), intr[name])).numpy()
    cu = out["colour"].transpose(1, 2, 0)
    # same scene, same camera: the two renderers differ only by the alpha cut-offs / radius rule / tiling
    assert np.abs(img - cu).mean() < 3e-3
    assert np.abs(img[:, ::-1] - cu).mean() > 0.05           # and the python image is the horizontally flipped convention
    # invariants of the spec
    c, pix = out["contrib"], out["pixels"]
    assert c.min() >= 0.0 and c.max() <= 0.99 + 1e-6
    assert ((pix >= 0) & (pix < 320 * 180)).all()
    seen = c > 0
    assert (out["radii"][seen] > 0).all()
    assert (out["surf"][seen] < 3e38).all()                  # a blended Gaussian always gets a surface distance
    assert np.isfinite(out["depth"]).all() and out["depth"].min() >= 0.0
    # masking every pixel silences everything
    O2 = ref_cuda.CudaRasterizerOracle(sc.xyz.numpy(), sc.opacities.numpy(), RG.strip_symmetric(cov).numpy(),
                                       colors_precomp=sc.colours.numpy())
    out2 = O2.forward(ref_cuda.camera_settings(tr[name], intr[name]), mask=np.zeros(320 * 180, np.int32))
    assert out2["contrib"].max() == 0.0 and out2["colour"].max() == 0.0
