"""
Drop-in mirror of the reference's native package ``gaussian_pointcloud_rasterization``
(gaussian-pointcloud-rasterization/gaussian_pointcloud_rasterization/__init__.py): ``GaussianRasterizationSettings``
and ``GaussianRasterizer`` with the same constructor contract, ``forward(raster_settings)`` result
``(colour[3,H,W], radii, invdepths, depths)``, running state and getters -- on the HIP rasteriser in libg2pc.so
(``g2pc_raster_front_cu`` / ``g2pc_raster_back_cu``, native-rasteriser semantics of SURVEY.md §8(a.5)).
The binding-side reductions of the reference (colour gather at the arg-max pixel, strict-> running max, running sum,
running min; __init__.py:128-158) are fused into the last kernel of the camera.
"""
import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from g2pc import _native as nv

FLT_MAX = float(torch.finfo(torch.float).max)


class GaussianRasterizationSettings(NamedTuple):
    """__init__.py:21-35 (same 14 fields, same order)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    mask: Optional[torch.Tensor]
    prefiltered: bool
    debug: bool
    antialiasing: bool


class _Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("focal_x", C.c_float), ("focal_y", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("bg", C.c_float * 3)]


_vp = C.c_void_p
nv._RASTER_PROTOS.update({
    "g2pc_raster_front_workspace": (C.c_size_t, [C.c_int64]),
    "g2pc_raster_back_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "g2pc_raster_front_cu": (C.c_int, [C.POINTER(_Camera)] + [_vp] * 5 + [C.c_int32, C.c_int32, _vp, C.c_int64] +
                             [_vp] * 8 + [C.c_size_t, _vp]),
    "g2pc_raster_back_cu": (C.c_int, [C.POINTER(_Camera), _vp, C.c_int64, C.c_int64] + [_vp] * 6 + [C.c_int] + [_vp] * 10 +
                            [C.c_int32] + [_vp] * 4 + [C.c_size_t, _vp]),
})
if nv._LIB is not None:
    nv._bind(nv._LIB)


class GaussianRasterizer(nn.Module):
    def __init__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, visible_gaussian_threshold=0.0, surface_distance_std=None,
                 calculate_surface_distance=False):
        super().__init__()

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        nv.lib()
        dev = means3D.device
        self.device = dev
        n = means3D.shape[0]
        self.n = n
        self.means3D = means3D.to(torch.float32).contiguous()
        self.means2D = means2D
        self.opacities = opacities.to(torch.float32).reshape(-1).contiguous()
        self.shs = shs.to(torch.float32).contiguous() if shs is not None else None
        self.colors_precomp = colors_precomp.to(torch.float32).contiguous() if colors_precomp is not None else None
        if cov3D_precomp is None:
            # computeCov3D (forward.cu:116-150): Sigma from ACTIVATED scales and the quaternion as given -- the same
            # product the fused covariance kernel forms from log-scales
            from g2pc import ops
            cov3D_precomp = ops.build_covariances(torch.log(scales.to(torch.float32)), rotations, 1.0, want_cov6=True)[1]
        self.cov3D_precomp = cov3D_precomp.to(torch.float32).contiguous()

        # running state (__init__.py:60-72)
        self.gaussian_max_contribution = torch.zeros(n, device=dev, dtype=torch.float)
        self.gaussian_min_surface_distance = torch.full((n,), FLT_MAX, device=dev, dtype=torch.float)
        self.gaussian_total_contribution = torch.zeros(n, device=dev, dtype=torch.float)
        self.gaussian_colours = torch.zeros((n, 3), device=dev, dtype=torch.float)

        self.visible_gaussian_threshold = visible_gaussian_threshold
        self.surface_distance_std = surface_distance_std
        self.calculate_surface_distance = calculate_surface_distance

        # per-camera scratch, allocated once
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self._p0, self._p1 = torch.empty((n, 4), **f32), torch.empty((n, 4), **f32)
        self._rect, self._sorted, self._offsets = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32)
        self._rgb = torch.empty((n, 3), **f32)
        self._cam_key = torch.empty(n, dtype=torch.int64, device=dev)
        self._cam_surf = torch.empty(n, **i32)
        self._winner_cam = torch.full((n,), 1 << 30, **i32)
        self._camera_counter = 0
        self._front_bytes = nv.lib().g2pc_raster_front_workspace(n)
        self._front_ws = nv.workspace(self._front_bytes, dev)
        self._back_bytes, self._back_ws = 0, None
        self.last = {}

    def forward(self, raster_settings, return_per_camera=False, cam_index=None):
        """__init__.py:90-140.  cam_index: global camera order (multi-GPU camera sharding); default = call order."""
        if cam_index is None:
            cam_index = self._camera_counter
        self._camera_counter = int(cam_index) + 1
        L = nv.lib()
        rs = raster_settings
        st = nv.stream_handle(self.device)
        H, W = int(rs.image_height), int(rs.image_width)
        cam = _Camera()
        cam.view[:] = rs.viewmatrix.reshape(-1).tolist()
        cam.proj[:] = rs.projmatrix.reshape(-1).tolist()
        cam.tan_fovx, cam.tan_fovy = rs.tanfovx, rs.tanfovy
        cam.width, cam.height = W, H
        cam.bg[:] = rs.bg.reshape(-1).tolist()
        campos = (C.c_float * 3)(*rs.campos.reshape(-1).tolist())
        mask = rs.mask.to(device=self.device, dtype=torch.int32).contiguous() if rs.mask is not None else None
        n = self.n
        radii = torch.empty(n, dtype=torch.int32, device=self.device)
        shs = self.shs
        coeffs = int(shs.shape[1]) if shs is not None else 0
        with nv.region("raster_front", self.device):
            nv.check(L.g2pc_raster_front_cu(C.byref(cam), nv.ptr(self.means3D), nv.ptr(self.cov3D_precomp),
                                            nv.ptr(self.opacities), nv.ptr(self.colors_precomp), nv.ptr(shs),
                                            int(rs.sh_degree) if shs is not None else 0, coeffs,
                                            C.cast(campos, C.c_void_p), n, nv.ptr(self._p0), nv.ptr(self._p1),
                                            nv.ptr(self._rect), nv.ptr(self._rgb), nv.ptr(radii), nv.ptr(self._sorted),
                                            nv.ptr(self._offsets), nv.ptr(self._front_ws), self._front_bytes, st),
                     "raster_front_cu")
        num_rendered = int(self._offsets[n].item())              # rasterizer_impl.cu:289 has the same read-back
        tiles_n = ((W + 15) // 16) * ((H + 15) // 16)
        need = L.g2pc_raster_back_workspace(num_rendered, tiles_n)
        if need > self._back_bytes:
            self._back_bytes = int(need * 1.25)
            self._back_ws = nv.workspace(self._back_bytes, self.device)
        colour = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
        depths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
        invdepths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
        cur = (torch.empty(n, dtype=torch.float32, device=self.device), torch.empty(n, dtype=torch.int32, device=self.device),
               torch.empty(n, dtype=torch.float32, device=self.device)) if return_per_camera else (None, None, None)
        with nv.region("raster_back_cu", self.device):
            nv.check(L.g2pc_raster_back_cu(C.byref(cam), nv.ptr(mask), n, num_rendered, nv.ptr(self._p0), nv.ptr(self._p1),
                                           nv.ptr(self._rect), nv.ptr(self._rgb), nv.ptr(self._sorted),
                                           nv.ptr(self._offsets), 1 if self.calculate_surface_distance else 0,
                                           nv.ptr(self._cam_key), nv.ptr(self._cam_surf), nv.ptr(colour), nv.ptr(depths),
                                           nv.ptr(invdepths), nv.ptr(self.gaussian_max_contribution),
                                           nv.ptr(self.gaussian_total_contribution), nv.ptr(self.gaussian_colours),
                                           nv.ptr(self.gaussian_min_surface_distance), nv.ptr(self._winner_cam),
                                           int(cam_index), nv.ptr(cur[0]), nv.ptr(cur[1]), nv.ptr(cur[2]),
                                           nv.ptr(self._back_ws), self._back_bytes, st),
                     "raster_back_cu")
        self.last = dict(num_rendered=num_rendered, contributions=cur[0], pixels=cur[1], surface_distances=cur[2])
        return colour, radii, invdepths, depths

    # the reference's renderer objects are called like functions by the pipeline (gauss_to_pc.py:454)
    def __call__(self, raster_settings, **kwargs):
        kwargs.pop("return_image", None)
        slot = kwargs.pop("slot", None)
        if slot is not None:
            kwargs["cam_index"] = slot - 1
        return self.forward(raster_settings, **kwargs)

    # ---- getters (__init__.py:160-219) ----------------------------------------------------------------------
    def get_gaussian_colours(self):
        return self.gaussian_colours * 255

    def get_max_gaussian_contributions(self):
        return self.gaussian_max_contribution

    def get_total_gaussian_contributions(self):
        return self.gaussian_total_contribution

    def get_gaussians_above_contribution_threshold(self, contribution_threshold):
        return self.get_max_gaussian_contributions() > contribution_threshold

    def get_gaussians_above_total_contribution_threshold(self, contribution_threshold):
        return self.get_total_gaussian_contributions() > contribution_threshold

    def get_surface_gaussians_below_distance_threshold(self, surface_distance_threshold):
        if not self.calculate_surface_distance:
            raise Exception("Cannot determine Gaussian surface distance as this feature was not set at the start of rendering")
        surface_indices = (self.gaussian_min_surface_distance < FLT_MAX)
        mean_and_std = torch.std_mean(self.gaussian_min_surface_distance[surface_indices])
        return self.gaussian_min_surface_distance < mean_and_std[1] * surface_distance_threshold

    def get_visible_gaussians(self):
        return self.get_gaussians_above_contribution_threshold(self.visible_gaussian_threshold)

    def get_gaussians_with_low_surface_distance(self):
        return self.get_surface_gaussians_below_distance_threshold(self.surface_distance_std)

    def get_predicted_surface_gaussians(self, predicted_surface_std=0.5):
        return self.get_surface_gaussians_below_distance_threshold(predicted_surface_std)

    def all_reduce_visibility(self, group=None):
        """Multi-GPU, cameras sharded over ranks.  max / min / sum are exact up to the fp32 order of the SUM; the
        winner's colour is selected by (max contribution, then earliest global camera index)."""
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        gmax = self.gaussian_max_contribution.clone()
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
        # earliest camera among the ranks that reached the global maximum (strict > in the reference = first wins)
        none = torch.full_like(self._winner_cam, 1 << 30)
        cand = torch.where((self.gaussian_max_contribution == gmax) & (gmax > 0), self._winner_cam, none)
        first = cand.clone()
        dist.all_reduce(first, op=dist.ReduceOp.MIN, group=group)
        mine = (cand == first) & (first < (1 << 30))
        self.gaussian_colours = torch.where(mine.unsqueeze(1), self.gaussian_colours, torch.zeros_like(self.gaussian_colours))
        dist.all_reduce(self.gaussian_colours, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.gaussian_total_contribution, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.gaussian_min_surface_distance, op=dist.ReduceOp.MIN, group=group)
        self._winner_cam = first
        self.gaussian_max_contribution = gmax

    def rebase_keys(self):
        pass
