"""
``gaussian_pointcloud_rasterization._C`` -- the reference's native extension module, on the HIP library.

The reference binds its CUDA rasteriser as a pybind11 module with two functions (ext.cpp:15-18):

    rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, mask, prefiltered,
                        antialiasing, calculate_surface_distance, debug)
        -> (num_rendered, out_color, out_depth, radii, geomBuffer, binningBuffer, imgBuffer, out_invdepth,
            gauss_contributions, gauss_surface_distances, gauss_pixels)            rasterize_points.h:18-41, .cu:36-145
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]                       rasterize_points.h:43-46, .cu:147-166

This module has the same two functions with the same 22 / 3 positional arguments and the same results, over
``g2pc_rasterize_gaussians`` / ``g2pc_mark_visible`` of libg2pc.so (include/g2pc.h).  The reference's UNMODIFIED binding
(gaussian_pointcloud_rasterization/__init__.py:90-158: ``from . import _C`` ... ``_C.rasterize_gaussians(*args)``) runs on it
as it is: drop this file next to it in place of the compiled ``_C*.so`` (tests/test_c_entry.py does
exactly that, in the container against the reference's own CUDA sources compiled for the host, and on the MI355X against
their stored results).

Like the reference's entry it blocks once per call to read the instance count (rasterizer_impl.cu:289).  The production
pipeline of this package does not go through here (``GaussianRasterizer.forward`` of the drop-in ``__init__.py`` keeps the
cameras in flight without the host in the loop); this module is the boundary a maintainer of the reference binds.
"""
import ctypes as C

import torch

from g2pc import _native as nv

_RESIZE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class _Args(C.Structure):
    _fields_ = [("background", C.c_void_p), ("means3D", C.c_void_p), ("colors", C.c_void_p), ("opacity", C.c_void_p),
                ("scales", C.c_void_p), ("rotations", C.c_void_p), ("scale_modifier", C.c_float), ("cov3D_precomp", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("image_height", C.c_int32), ("image_width", C.c_int32), ("sh", C.c_void_p), ("degree", C.c_int32),
                ("campos", C.c_void_p), ("mask", C.c_void_p), ("prefiltered", C.c_int32), ("antialiasing", C.c_int32),
                ("calculate_surface_distance", C.c_int32), ("debug", C.c_int32), ("P", C.c_int64), ("M", C.c_int32)]


class _Out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("out_color", "out_depth", "radii", "out_invdepth", "gauss_contributions",
                                          "gauss_surface_distances", "gauss_pixels")]


nv._RASTER_PROTOS.update({
    "g2pc_rasterize_gaussians": (C.c_int, [C.POINTER(_Args), C.POINTER(_Out), C.POINTER(C.c_int32), _RESIZE, C.c_void_p, _RESIZE,
                                           C.c_void_p, _RESIZE, C.c_void_p, C.c_void_p]),
    "g2pc_mark_visible": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_float * 16), C.c_void_p, C.c_void_p]),
})
if nv._LIB is not None:
    nv._bind(nv._LIB)


def _host(t, n):
    """The small tensors that become kernel arguments (background, matrices, camera position): n floats on the host."""
    v = t.detach().to(device="cpu", dtype=torch.float32).reshape(-1)
    if v.numel() != n:
        raise RuntimeError("expected %d values, got a tensor of shape %s" % (n, tuple(t.shape)))
    return (C.c_float * n)(*v.tolist())


def _dev(t, device):
    """Device pointer of a float32 / int32 tensor; an EMPTY tensor is the reference's "absent" (null data pointer)."""
    if t is None or t.numel() == 0:
        return None, None
    t = t.to(device=device).contiguous()
    return t, nv.ptr(t)


class _Buffer:
    """resizeFunctional (rasterize_points.cu:25-34): a byte tensor the library sizes through a callback."""

    def __init__(self, device):
        self.t = torch.empty((0,), dtype=torch.uint8, device=device)

        def resize(_user, nbytes):
            self.t.resize_((int(nbytes),))
            return self.t.data_ptr()

        self.fn = _RESIZE(resize)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, mask, prefiltered,
                        antialiasing, calculate_surface_distance, debug):
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")          # rasterize_points.cu:61-63
    L = nv.lib()
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
    # rasterize_points.cu:73-93 (the library writes the initial values itself; P == 0 leaves the images zero)
    out_color = torch.empty((3, H, W), **f32)
    out_depth = torch.empty((1, H, W), **f32)
    out_invdepth = torch.empty((1, H, W), **f32)
    radii = torch.zeros((P,), **i32)
    contrib = torch.zeros((P,), **f32)
    surf = torch.full((P,), float(torch.finfo(torch.float32).max), **f32)
    pixels = torch.zeros((P,), **i32)
    geom, binning, img = _Buffer(dev), _Buffer(dev), _Buffer(dev)
    keep = []

    def d(t, dtype=torch.float32):
        t, p = _dev(t if t is None else t.to(dtype), dev)
        keep.append(t)
        return p

    a = _Args()
    bg, vm, pm, cp = _host(background, 3), _host(viewmatrix, 16), _host(projmatrix, 16), _host(campos, 3)
    a.background, a.viewmatrix, a.projmatrix, a.campos = (C.cast(x, C.c_void_p) for x in (bg, vm, pm, cp))
    a.means3D, a.colors, a.opacity = d(means3D), d(colors), d(opacity.reshape(-1))
    a.scales, a.rotations, a.cov3D_precomp, a.sh = d(scales), d(rotations), d(cov3D_precomp), d(sh)
    a.scale_modifier, a.tan_fovx, a.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
    a.image_height, a.image_width, a.degree = H, W, int(degree)
    a.mask = d(mask.reshape(-1) if mask is not None else None, torch.int32)
    a.prefiltered, a.antialiasing = int(bool(prefiltered)), int(bool(antialiasing))
    a.calculate_surface_distance, a.debug = int(bool(calculate_surface_distance)), int(bool(debug))
    a.P, a.M = P, (int(sh.shape[1]) if sh is not None and sh.numel() else 0)
    o = _Out(*[nv.ptr(t) if t.numel() else None for t in (out_color, out_depth, radii, out_invdepth, contrib, surf, pixels)])
    rendered = C.c_int32(0)
    rc = L.g2pc_rasterize_gaussians(C.byref(a), C.byref(o), C.byref(rendered), geom.fn, None, binning.fn, None, img.fn, None,
                                    nv.stream_handle(dev))
    if rc != 0:
        msg = L.g2pc_last_error()
        raise RuntimeError("rasterize_gaussians failed (%d): %s" % (rc, msg.decode() if msg else ""))   # std::runtime_error there
    return (int(rendered.value), out_color, out_depth, radii, geom.t, binning.t, img.t, out_invdepth, contrib, surf, pixels)


def mark_visible(means3D, viewmatrix, projmatrix):
    """rasterize_points.cu:147-166 -> checkFrustum -> in_frustum (auxiliary.h:151-176): z_view > 0.2.  The reference computes
    p_proj from projmatrix there and never uses it."""
    L = nv.lib()
    P = int(means3D.shape[0])
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        pos = means3D.to(torch.float32).contiguous()
        out = torch.empty((P,), dtype=torch.uint8, device=pos.device)
        view = _host(viewmatrix, 16)
        nv.check(L.g2pc_mark_visible(nv.ptr(pos), P, C.byref(view), nv.ptr(out), nv.stream_handle(pos.device)), "mark_visible")
        present = out.bool()
    return present
