"""oracle/ref_render.py pinned against the untouched reference's outputs (tests/golden/render_py_n6000.npz)."""
import os

import numpy as np
import torch

import ref_gauss as RG
import ref_render as RR
from g2pc.synth import make_scene, make_cameras


def test_python_renderer_oracle_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "render_py_n6000.npz"))
    sc = make_scene(int(g["n"]), int(g["seed"]), scale_lo=float(g["scale_lo"]), scale_hi=float(g["scale_hi"]))
    transforms, intr = make_cameras(int(g["ncam"]), width=int(g["width"]), height=int(g["height"]), focal=float(g["focal"]))
    cov = RG.covariances(sc.scales, sc.rots)
    R = RR.PythonRendererOracle(sc.xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05)
    for i, name in enumerate(transforms):
        cam = RR.get_camera(torch.tensor(transforms[name]), intr[name])
        img = R(cam)
        assert np.array_equal(img.numpy().astype(np.float32), g["images"][i])          # bit-exact on CPU
        assert np.array_equal(R.max_contribution.numpy(), g["contrib_after_cam"][i])
    assert np.array_equal(R.get_gaussian_colours().numpy(), g["colours"])
    assert np.array_equal(R.get_visible_gaussians().numpy(), g["visible"])


def _split_case(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "render_py_split_%s.npz" % tag))
    sc = make_scene(int(g["n"]), int(g["seed"]), scale_lo=float(g["scale_lo"]), scale_hi=float(g["scale_hi"]))
    xyz = sc.xyz * float(g["crowd"])
    transforms, intr = make_cameras(1, width=int(g["width"]), height=int(g["height"]), focal=float(g["focal"]))
    name = next(iter(transforms))
    pin = int(g["tile_pin"])
    return g, sc, xyz, torch.tensor(transforms[name]), intr[name], dict(max_gaussians_per_tile=pin, max_tile_size=pin // 1000)


def test_oracle_splits_overloaded_leaves_like_the_reference(golden_dir):
    """The count-driven split of the reference's queue (gauss_render.py:319-335), from outputs of the untouched reference
    (oracle/make_golden.py render_split): `60k` = leaves over the pinned default of 60 000 Gaussians, `deep` = several levels
    down to dropped children under a 10 000 / 10-pixel limit."""
    for tag in ("60k", "deep"):
        g, sc, xyz, c2w, intr, limits = _split_case(golden_dir, tag)
        assert int(g["splits"]) > {"60k": 5, "deep": 21}[tag]                      # more splits than the image size alone causes
        cov = RG.covariances(sc.scales, sc.rots)
        R = RR.PythonRendererOracle(xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05, **limits)
        img = R(RR.get_camera(c2w, intr))
        assert np.array_equal(img.numpy().astype(np.float32), g["image"]), tag
        assert np.array_equal(R.max_contribution.numpy(), g["contrib"]), tag
        cols = R.get_gaussian_colours().numpy()
        assert np.array_equal(cols[::int(g["stride"])], g["colours"]) and np.allclose(cols.sum(axis=0), g["colour_sum"], rtol=1e-12), tag
        assert np.array_equal(np.packbits(R.get_visible_gaussians().numpy()), g["visible"]), tag
