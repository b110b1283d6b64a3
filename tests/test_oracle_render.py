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
