"""
TEST INFRASTRUCTURE ONLY (oracle) -- python side of the native-rasteriser restatement: loads
oracle/_build/libcuda_raster_ref.so (built from oracle/cuda_raster_ref.c by build_oracle()) and mirrors the
reference's binding class (gaussian_pointcloud_rasterization/__init__.py:38-219): per-camera forward + the
running max-contribution / colour / total-contribution / min-surface-distance state and its getters.
PARITY UNPINNED (see the header of cuda_raster_ref.c).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libcuda_raster_ref.so")
FLT_MAX = float(np.finfo(np.float32).max)


def build_oracle() -> str:
    src = os.path.join(HERE, "cuda_raster_ref.c")
    if not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", SO, src, "-lm"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.cuda_ref_forward.restype = C.c_int
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def camera_settings(c2w, intr, colour_resolution=None, white_bkgd=True):
    """camera_handler.py:53-108 (renderer_type="cuda"): flips columns 1:3 of c2w, view = inv(c2w)^T, proj = view @ P^T."""
    w0, h0, fx0, fy0 = intr
    diff = 1 if colour_resolution is None else colour_resolution / int(w0)
    W, H = int(int(w0) * diff), int(int(h0) * diff)
    fx, fy = float(fx0) * diff, float(fy0) * diff
    t = np.array(c2w, dtype=np.float32).copy()
    t[:, 1:3] = -t[:, 1:3]
    fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
    tfx, tfy = math.tan(fovx * 0.5), math.tan(fovy * 0.5)
    znear, zfar = 10, 100
    P = np.zeros((4, 4), dtype=np.float32)
    top, right = tfy * znear, tfx * znear
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    view = np.linalg.inv(t).T.astype(np.float32)
    campos = np.linalg.inv(view)[3, :3].astype(np.float32)
    return dict(H=H, W=W, tanfovx=tfx, tanfovy=tfy, viewmatrix=view, projmatrix=(view @ P.T).astype(np.float32),
                campos=campos, bg=np.array([1.0, 1.0, 1.0] if white_bkgd else [0.0, 0.0, 0.0], dtype=np.float32))


class CudaRasterizerOracle:
    def __init__(self, means3D, opacities, cov6, colors_precomp=None, shs=None, sh_degree=3, threshold=0.0,
                 surface_distance_std=None, calculate_surface_distance=False):
        self.means, self.opac, self.cov6 = _f(means3D), _f(opacities).reshape(-1), _f(cov6)
        self.colors, self.shs, self.deg = _f(colors_precomp), _f(shs), sh_degree
        n = self.means.shape[0]
        self.n = n
        self.max_contribution = np.zeros(n, np.float32)
        self.min_surface = np.full(n, FLT_MAX, np.float32)
        self.total = np.zeros(n, np.float32)
        self.colours = np.zeros((n, 3), np.float32)
        self.threshold, self.surf_std, self.calc_surf = threshold, surface_distance_std, calculate_surface_distance

    def forward(self, cam, mask=None):
        H, W, n = cam["H"], cam["W"], self.n
        mask = np.ones(H * W, np.int32) if mask is None else np.ascontiguousarray(mask, np.int32).reshape(-1)
        out_color = np.zeros((3, H, W), np.float32)
        out_depth, out_inv = np.zeros((1, H, W), np.float32), np.zeros((1, H, W), np.float32)
        radii, pix = np.zeros(n, np.int32), np.zeros(n, np.int32)
        contrib, surf = np.zeros(n, np.float32), np.zeros(n, np.float32)
        M = self.shs.shape[1] if self.shs is not None else 0
        L = lib().cuda_ref_forward(
            C.c_int(n), C.c_int(self.deg), C.c_int(M), _p(cam["bg"]), _p(self.means), _p(self.colors), _p(self.opac),
            _p(self.cov6), _p(np.ascontiguousarray(cam["viewmatrix"])), _p(np.ascontiguousarray(cam["projmatrix"])),
            C.c_float(cam["tanfovx"]), C.c_float(cam["tanfovy"]), C.c_int(H), C.c_int(W), _p(self.shs), _p(cam["campos"]),
            _p(mask), C.c_int(1 if self.calc_surf else 0), _p(out_color), _p(out_depth), _p(out_inv), _p(radii),
            _p(contrib), _p(surf), _p(pix))
        flat = out_color.transpose(1, 2, 0).reshape(-1, 3)
        cols = flat[pix]
        upd = contrib > self.max_contribution                        # __init__.py:142-152
        self.max_contribution[upd] = contrib[upd]
        self.colours[upd] = cols[upd]
        self.total += contrib
        lower = surf < self.min_surface                              # :154-158
        self.min_surface[lower] = surf[lower]
        return dict(colour=out_color, radii=radii, invdepth=out_inv, depth=out_depth, contrib=contrib, surf=surf,
                    pixels=pix, num_rendered=L)

    def get_gaussian_colours(self):
        return self.colours * 255

    def get_visible_gaussians(self):
        return self.max_contribution > self.threshold

    def get_total_gaussian_contributions(self):
        return self.total

    def get_surface_gaussians_below_distance_threshold(self, k):
        valid = self.min_surface < FLT_MAX
        mean = self.min_surface[valid].astype(np.float32).mean(dtype=np.float32)     # std_mean(...)[1] is the MEAN
        return self.min_surface < mean * np.float32(k)
