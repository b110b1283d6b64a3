"""Checks of the reference-shaped native entry gaussian_pointcloud_rasterization._C (g2pc_rasterize_gaussians, ABI 7).

drive_fixture:     the 22-argument call itself against the golden vectors of the reference's own rasteriser, with the binding's
                   reductions (gaussian_pointcloud_rasterization/__init__.py:128-158) restated here in numpy -- runs anywhere
                   (emulator in the CPU suite, the MI355X in the GPU suite).
reference_binding: the reference's UNMODIFIED __init__.py imported over this package's _C.py -- the one-line swap of
                   INTEGRATION.md section 4 (container only: needs /root/reference)."""
import importlib.util
import os
import sys

import numpy as np
import torch

import cu_golden

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3dgs-to-pc_amd")
GPR_NAME = "gaussian_pointcloud_rasterization"
FLT_MAX = float(np.finfo(np.float32).max)


def load_c_module():
    """This package's gaussian_pointcloud_rasterization._C WITHOUT the drop-in __init__.py around it."""
    spec = importlib.util.spec_from_file_location("_g2pc_C_entry", os.path.join(PKG, GPR_NAME, "_C.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _case_inputs(case, dev):
    from gauss_handler import Gaussians
    sc, transforms, intr = case.scene()
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    c6 = torch.from_numpy(case.cov6(G.covariances.reshape(-1, 9)[:, [0, 1, 2, 4, 5, 8]].cpu().numpy())).to(dev)
    return sc, transforms, c6


def drive_fixture(name, device="cpu"):
    """Every camera of fixture `name` through _C.rasterize_gaussians with the arguments the reference's binding passes
    (__init__.py:100-123), then the binding's state updates in numpy; returns (camera reports, state report, case)."""
    _C = load_c_module()
    case = cu_golden.Case(name)
    r = case.recipe
    dev = torch.device(device)
    sc, transforms, c6 = _case_inputs(case, dev)
    n, W, H = case.n, case.W, case.H
    empty = torch.Tensor([])
    shs = sc.shs.to(dev) if r["with_sh"] else empty
    colours = empty if r["with_sh"] else sc.colours.to(dev)
    smax, stot = np.zeros(n, np.float32), np.zeros(n, np.float32)
    smin, scol = np.full(n, FLT_MAX, np.float32), np.zeros((n, 3), np.float32)
    reps = []
    for i in range(case.ncam):
        mask = torch.from_numpy(case.mask.reshape(-1).copy()).to(dev) if case.has_mask else \
            torch.full((H * W,), 1, dtype=torch.int32, device=dev)                                   # __init__.py:95-98
        out = _C.rasterize_gaussians(
            torch.tensor([1.0, 1.0, 1.0]), sc.xyz.to(dev), colours, sc.opacities.unsqueeze(1).to(dev), empty, empty, 1.0, c6,
            torch.from_numpy(case.cam(i, "viewmatrix")), torch.from_numpy(case.cam(i, "projmatrix")),
            float(case.cam(i, "tanfovx")), float(case.cam(i, "tanfovy")), H, W, shs, 3, torch.from_numpy(case.cam(i, "campos")),
            mask, False, False, bool(r["surf"]), True)
        assert len(out) == 11
        num_rendered, colour, depths, radii, geom, binning, img, invd, contrib, surf, pixels = out
        assert isinstance(num_rendered, int) and colour.shape == (3, H, W) and depths.shape == (1, H, W) and invd.shape == (1, H, W)
        assert radii.dtype == torch.int32 and pixels.dtype == torch.int32 and contrib.dtype == torch.float32
        assert geom.dtype == torch.uint8 and binning.dtype == torch.uint8 and img.dtype == torch.uint8
        assert geom.numel() > 0 and (binning.numel() > 0 or num_rendered == 0)
        got = dict(radii=radii.cpu().numpy(), num_rendered=num_rendered, out_color=colour.cpu().numpy(),
                   out_depth=depths.cpu().numpy(), out_invdepth=invd.cpu().numpy(), gauss_contributions=contrib.cpu().numpy(),
                   gauss_pixels=pixels.cpu().numpy(), gauss_surface_distances=surf.cpu().numpy())
        reps.append(cu_golden.compare_camera(case, i, got))
        # the binding's reductions (__init__.py:128-158), restated
        flat = got["out_color"].reshape(3, -1).T
        newcol = flat[got["gauss_pixels"].astype(np.int64)]
        c = got["gauss_contributions"]
        upd = c > smax
        smax[upd], scol[upd] = c[upd], newcol[upd]
        stot += c
        smin = np.minimum(smin, got["gauss_surface_distances"])
    st = dict(max_contribution=smax, total_contribution=stot, min_surface_distance=smin, colours=scol * 255.0, visible=smax > 0.05)
    if r["surf"]:
        reached = smin < FLT_MAX
        mean = smin[reached].mean(dtype=np.float32) if reached.any() else np.float32(0)
        st["low_surface_distance"] = smin < mean * np.float32(2.0)
        st["predicted_surface"] = smin < mean * np.float32(0.5)
    return reps, cu_golden.compare_state(case, st), case


def load_reference_binding_over(c_dir):
    """The reference's gaussian_pointcloud_rasterization/__init__.py, UNMODIFIED, with `from . import _C` resolving inside
    `c_dir` (this package's directory -> _C.py on libg2pc; oracle/_ref/<variant> -> the reference's own sources compiled for
    the host).  Not left in sys.modules."""
    import ref_shim
    init = os.path.join(ref_shim.REFERENCE_ROOT, "gaussian-pointcloud-rasterization", GPR_NAME, "__init__.py")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == GPR_NAME or k.startswith(GPR_NAME + ".")}
    try:
        spec = importlib.util.spec_from_file_location(GPR_NAME, init, submodule_search_locations=[c_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[GPR_NAME] = mod
        with ref_shim.CudaToCpu():
            spec.loader.exec_module(mod)
        assert os.path.dirname(os.path.abspath(mod._C.__file__)) == os.path.abspath(c_dir), mod._C.__file__
    finally:
        for k in [k for k in sys.modules if k == GPR_NAME or k.startswith(GPR_NAME + ".")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return mod


def run_reference_binding(mod, case, with_scales=False, antialiasing=False, cams=None):
    """Drive `mod.GaussianRasterizer` (the reference's class) over the fixture's cameras; returns its running state and the last
    camera's images.  with_scales: scale / rotation pair instead of cov3D_precomp (computeCov3D inside the rasteriser)."""
    import ref_shim
    sc, transforms, c6 = _case_inputs(case, torch.device("cpu"))
    r = case.recipe
    with ref_shim.CudaToCpu():
        kw = dict(shs=sc.shs) if r["with_sh"] else dict(colors_precomp=sc.colours)
        if with_scales:
            kw.update(scales=torch.exp(sc.scales), rotations=sc.rots)
        else:
            kw.update(cov3D_precomp=c6)
        R = mod.GaussianRasterizer(sc.xyz, torch.zeros_like(sc.xyz), sc.opacities.unsqueeze(1), visible_gaussian_threshold=0.05,
                                   surface_distance_std=2.0 if r["surf"] else None, calculate_surface_distance=bool(r["surf"]), **kw)
        last = None
        for i in (range(case.ncam) if cams is None else cams):
            rs = mod.GaussianRasterizationSettings(
                image_height=case.H, image_width=case.W, tanfovx=float(case.cam(i, "tanfovx")), tanfovy=float(case.cam(i, "tanfovy")),
                bg=torch.tensor([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=torch.from_numpy(case.cam(i, "viewmatrix")),
                projmatrix=torch.from_numpy(case.cam(i, "projmatrix")), sh_degree=3, campos=torch.from_numpy(case.cam(i, "campos")),
                mask=torch.from_numpy(case.mask.reshape(-1).copy()) if case.has_mask else None, prefiltered=False, debug=True,
                antialiasing=antialiasing)
            last = R.forward(rs)
        out = dict(max=R.gaussian_max_contribution.numpy().copy(), total=R.gaussian_total_contribution.numpy().copy(),
                   min_surf=R.gaussian_min_surface_distance.numpy().copy(), colours=R.get_gaussian_colours().numpy().copy(),
                   visible=R.get_visible_gaussians().numpy().copy(), image=last[0].numpy().copy(), radii=last[1].numpy().copy(),
                   invdepth=last[2].numpy().copy(), depth=last[3].numpy().copy())
        if r["surf"]:
            out["low_surface"] = R.get_gaussians_with_low_surface_distance().numpy().copy()
    return out
