"""oracle/cuda_raster_ref.c (native-rasteriser restatement, parity UNPINNED -- the CUDA reference cannot be run here):
internal consistency checks that do not need the reference: it must agree with the (reference-pinned) python-renderer
oracle up to the documented semantic differences, and obey the invariants of the deterministic spec."""
import numpy as np
import torch

import ref_cuda
import ref_gauss as RG
import ref_render as RR
from g2pc.synth import make_scene, make_cameras


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
