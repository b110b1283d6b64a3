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
                ("bg", C.c_float * 3), ("lim_x", C.c_float), ("lim_y", C.c_float)]


_vp = C.c_void_p
nv._RASTER_PROTOS.update({
    "g2pc_raster_front_workspace": (C.c_size_t, [C.c_int64]),
    "g2pc_raster_back_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "g2pc_raster_front_cu": (C.c_int, [C.POINTER(_Camera)] + [_vp] * 5 + [C.c_int32, C.c_int32, _vp, C.c_int64] +
                             [_vp] * 7 + [C.c_size_t, _vp]),
    "g2pc_raster_back_cu": (C.c_int, [C.POINTER(_Camera), _vp, C.c_int64, C.c_int64] + [_vp] * 4 + [C.c_int] + [_vp] * 10 +
                            [C.c_int32] + [_vp] * 3 + [C.c_int, _vp, C.c_size_t, _vp]),
    "g2pc_raster_back_cu_tiles": (C.c_int, [C.POINTER(_Camera), _vp, C.c_int64, C.c_int64] + [_vp] * 4 + [C.c_int] + [_vp] * 10 +
                                  [C.c_int32] + [_vp] * 3 + [C.c_int, C.c_int32, C.c_int32, _vp, C.c_size_t, _vp]),
    "g2pc_raster_back_cu_dev": (C.c_int, [C.POINTER(_Camera), _vp, C.c_int64, C.c_int64] + [_vp] * 4 + [C.c_int] + [_vp] * 6 +
                                [C.c_int32, C.c_int32, _vp, C.c_size_t, _vp]),
    "g2pc_mark_visible": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_float * 16), _vp, _vp]),
    "g2pc_sh_planes": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp]),
    "g2pc_raster_camera_cu_workspace": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "g2pc_raster_camera_cu": (C.c_int, [C.POINTER(_Camera)] + [_vp] * 5 + [C.c_int32, C.c_int32, _vp, _vp, C.c_int64, C.c_int64] +
                              [_vp] * 3 + [C.c_int] + [_vp] * 6 + [C.c_int32, C.c_int32, _vp, C.c_size_t, _vp]),
})
if nv._LIB is not None:
    nv._bind(nv._LIB)


PIPELINE_STREAMS = 4
FUSED_CAMERA_CALL = True          # pipelined cameras through g2pc_raster_camera_cu (False: front + device-side back half, as until round 4)
PIPELINE_IN_EMULATOR = False      # tests: drive the no-read-back camera path through the CPU emulator too
CAPACITY_HEADROOM = 1.25          # instance capacity of the pipelined cameras relative to the largest count seen so far
MIN_CAPACITY = 1 << 16
SH_PLANES = True                  # SH coefficients transposed once per renderer to plane-major (g2pc_sh_planes): coalesced reads per camera


def mark_visible(means3D, viewmatrix, projmatrix=None):
    """_C.mark_visible(means3D, viewmatrix, projmatrix) -> bool[P] (rasterize_points.cu:147-166 -> checkFrustum ->
    in_frustum, auxiliary.h:151-176): the centre lies in front of the near plane, z_view > 0.2.  projmatrix is accepted
    for signature compatibility; the reference computes p_proj from it and never uses the result (auxiliary.h:160-166)."""
    positions = means3D.to(torch.float32).contiguous()
    out = torch.empty((positions.shape[0],), dtype=torch.uint8, device=positions.device)
    view = (C.c_float * 16)(*viewmatrix.reshape(-1).tolist())
    nv.check(nv.lib().g2pc_mark_visible(nv.ptr(positions), positions.shape[0], C.byref(view), nv.ptr(out),
                                        nv.stream_handle(positions.device)), "mark_visible")
    return out.bool()



class _CuScratch:
    """Per-stream scratch of one in-flight camera."""

    def __init__(self, n, dev, stream=None, pipelined=False):
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        self.rec = torch.empty((n, 16), **f32)                    # one 64-byte blend record per Gaussian
        # (rect: two words per Gaussian once the image exceeds 4096 pixels a side, include/g2pc.h g2pc_raster_front_cu)
        self.rect, self.sorted, self.offsets = torch.empty(2 * n, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32)
        self.radii = torch.empty(n, **i32)
        self.cam_key = torch.empty(n, dtype=torch.int64, device=dev)
        self.cam_surf = torch.empty(n, **i32)
        self.front_bytes = nv.lib().g2pc_raster_front_workspace(n)
        self.front_ws = nv.workspace(self.front_bytes, dev)
        self.back_bytes, self.back_ws = 0, None
        self.colour = self.depths = self.invdepths = None
        self.stream = stream
        # [instances, 0], written by the device through the pinned buffer's mapping -- pipelined scratches only: the
        # synchronous scratch reads offsets[n] back itself, a pageable buffer here would cost a blocking copy per camera
        self.count_host = torch.zeros((4,), dtype=torch.int32) if pipelined else None   # (instances, unsorted, -, -)
        if pipelined and stream is not None:
            self.count_host = self.count_host.pin_memory()
        self.front_done = torch.cuda.Event() if stream is not None else None
        self.update_done = torch.cuda.Event() if stream is not None else None


class GaussianRasterizer(nn.Module):
    needs_camera_epochs = False        # the winner camera is a plain int32 index here

    def __init__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, visible_gaussian_threshold=0.0, surface_distance_std=None,
                 calculate_surface_distance=False, tile_shard=None):
        super().__init__()
        # (rank, world): multi-GPU jobs with fewer cameras than ranks -- every rank renders every camera but blends only
        # tiles rank, rank + world, ...; the per-camera results are merged across ranks inside forward() (see _exchange_camera)
        self.tile_shard = tuple(tile_shard) if tile_shard is not None and tile_shard[1] > 1 else None
        self.tile_group = None

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

        self._sync = _CuScratch(n, dev)
        self._pipe, self._pipe_next, self._pending, self._last_update = [], 0, [], None
        self._capacity = None              # instance capacity of the pipelined cameras (learned from the first camera)
        self.rerendered = 0                # pipelined cameras that outgrew the capacity and went through the two-call path
        self._winner_cam = torch.full((n,), 1 << 30, dtype=torch.int32, device=dev)
        self._camera_counter = 0
        self.last = {}
        self._sh()          # (the plane-major copy of the SH coefficients is made HERE, on the constructing stream: the camera
        #                      streams are ordered behind it when the pipeline starts)

    # ---- one camera = front (async) / back (bin + blend) / ordered running-state update ------------------------------
    def _camera(self, rs):
        cam = _Camera()
        cam.view[:] = rs.viewmatrix.reshape(-1).tolist()
        cam.proj[:] = rs.projmatrix.reshape(-1).tolist()
        cam.tan_fovx, cam.tan_fovy = rs.tanfovx, rs.tanfovy
        cam.width, cam.height = int(rs.image_width), int(rs.image_height)
        cam.bg[:] = rs.bg.reshape(-1).tolist()
        campos = (C.c_float * 3)(*rs.campos.reshape(-1).tolist())
        mask = rs.mask.to(device=self.device, dtype=torch.int32).contiguous() if rs.mask is not None else None
        return cam, campos, mask

    def _stream_ptr(self, sc):
        return C.c_void_p(sc.stream.cuda_stream) if sc.stream is not None else nv.stream_handle(self.device)

    def _sh(self):
        """(coefficients, count) as the rasteriser calls take them: PLANE-MAJOR (g2pc_sh_planes, count < 0) once per renderer when the
        count is a multiple of 4 -- the scene is the same for every camera, and a wave then reads 1 KB contiguous per 16-byte
        vector instead of 64 pieces 192 bytes apart (k_preprocess_cu was bound by the address unit, not by HBM) --, else as given."""
        shs = self.shs
        if shs is None:
            return None, 0
        k = int(shs.shape[1])
        if not SH_PLANES or k % 4 != 0:
            return shs, k
        if getattr(self, "_sh_planes", None) is None:
            planes = torch.empty_like(shs)
            nv.check(nv.lib().g2pc_sh_planes(nv.ptr(shs), self.n, k, nv.ptr(planes), nv.stream_handle(self.device)), "sh_planes")
            self._sh_planes = planes
        return self._sh_planes, -k

    def _front(self, sc, cam, campos, sh_degree, count_host=True):
        shs, coeffs = self._sh()
        with nv.region("raster_front", self.device, sc.stream):
            nv.check(nv.lib().g2pc_raster_front_cu(
                C.byref(cam), nv.ptr(self.means3D), nv.ptr(self.cov3D_precomp), nv.ptr(self.opacities),
                nv.ptr(self.colors_precomp), nv.ptr(shs), int(sh_degree) if shs is not None else 0, coeffs,
                C.cast(campos, C.c_void_p), self.n, nv.ptr(sc.rec), nv.ptr(sc.rect), nv.ptr(sc.radii), nv.ptr(sc.sorted),
                nv.ptr(sc.offsets),
                C.c_void_p(sc.count_host.data_ptr()) if (count_host and sc.count_host is not None) else None, nv.ptr(sc.front_ws),
                sc.front_bytes, self._stream_ptr(sc)), "raster_front_cu")

    def _back(self, sc, cam, mask, num_rendered, cam_index, phases, name, cur=(None, None, None)):
        import contextlib
        L = nv.lib()
        W, H = cam.width, cam.height
        tiles_n = ((W + 15) // 16) * ((H + 15) // 16)
        need = L.g2pc_raster_back_workspace(num_rendered, tiles_n)
        with (torch.cuda.stream(sc.stream) if sc.stream is not None else contextlib.nullcontext()):
            if need > sc.back_bytes:
                sc.back_bytes = int(need * 1.25)
                sc.back_ws = nv.workspace(sc.back_bytes, self.device)
            if sc.colour is None or sc.colour.shape[1:] != (H, W):
                sc.colour = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
                sc.depths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
                sc.invdepths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
        with nv.region(name, self.device, sc.stream):
            first, step = self.tile_shard if self.tile_shard is not None else (0, 1)
            nv.check(L.g2pc_raster_back_cu_tiles(
                C.byref(cam), nv.ptr(mask), self.n, num_rendered, nv.ptr(sc.rec), nv.ptr(sc.rect), nv.ptr(sc.sorted),
                nv.ptr(sc.offsets), 1 if self.calculate_surface_distance else 0,
                nv.ptr(sc.cam_key), nv.ptr(sc.cam_surf), nv.ptr(sc.colour), nv.ptr(sc.depths), nv.ptr(sc.invdepths),
                nv.ptr(self.gaussian_max_contribution), nv.ptr(self.gaussian_total_contribution),
                nv.ptr(self.gaussian_colours), nv.ptr(self.gaussian_min_surface_distance), nv.ptr(self._winner_cam),
                int(cam_index), nv.ptr(cur[0]), nv.ptr(cur[1]), nv.ptr(cur[2]), phases, int(first), int(step),
                nv.ptr(sc.back_ws), sc.back_bytes, self._stream_ptr(sc)), "raster_back_cu")

    def _exchange_camera(self, sc):
        """Tile-split mode, between the blend and the running-state update of ONE camera: merge what the ranks found in their
        tiles -- MAX of the packed (contribution, ~pixel) keys, MIN of the surface distances, SUM of the images (every pixel
        was written by exactly one rank, zero elsewhere).  Afterwards the camera's data, and so the running state, is
        identical on every rank."""
        import torch.distributed as dist
        dist.all_reduce(sc.cam_key, op=dist.ReduceOp.MAX, group=self.tile_group)
        if self.calculate_surface_distance:
            dist.all_reduce(sc.cam_surf, op=dist.ReduceOp.MIN, group=self.tile_group)     # non-negative floats order like their bits
        dist.all_reduce(sc.colour, op=dist.ReduceOp.SUM, group=self.tile_group)
        dist.all_reduce(sc.depths, op=dist.ReduceOp.SUM, group=self.tile_group)
        dist.all_reduce(sc.invdepths, op=dist.ReduceOp.SUM, group=self.tile_group)

    def _issue(self, sc, cam, campos, mask, sh_degree, cam_index, capacity):
        """One pipelined camera, no host in the loop: front, then bin + blend sized for `capacity` with the instance count
        left on the device (g2pc_raster_back_cu_dev), then the running-state update in camera order."""
        import contextlib
        L = nv.lib()
        W, H = cam.width, cam.height
        tiles_n = ((W + 15) // 16) * ((H + 15) // 16)
        # ONE call per camera (g2pc_raster_camera_cu: the depth bucket sort emits the instances itself) where the library offers
        # it -- up to 256 x 256 tiles and ~2 M Gaussians --, else the front half + the device-side back half
        fused = FUSED_CAMERA_CALL and getattr(self, "_fused_ok", True) and max((W + 15) // 16, (H + 15) // 16) <= 256
        need = max(L.g2pc_raster_back_workspace(capacity, tiles_n),
                   L.g2pc_raster_camera_cu_workspace(self.n, capacity, tiles_n) if fused else 0)
        with (torch.cuda.stream(sc.stream) if sc.stream is not None else contextlib.nullcontext()):
            if need > sc.back_bytes:
                sc.back_bytes = int(need)
                sc.back_ws = None
                sc.back_ws = nv.workspace(sc.back_bytes, self.device)
            if sc.colour is None or sc.colour.shape[1:] != (H, W):
                sc.colour = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
                sc.depths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
                sc.invdepths = torch.empty((1, H, W), dtype=torch.float32, device=self.device)
        first, step = self.tile_shard if self.tile_shard is not None else (0, 1)
        if fused:
            shs, coeffs = self._sh()
            with nv.region("raster_camera_cu", self.device, sc.stream):
                rc = L.g2pc_raster_camera_cu(
                    C.byref(cam), nv.ptr(self.means3D), nv.ptr(self.cov3D_precomp), nv.ptr(self.opacities),
                    nv.ptr(self.colors_precomp), nv.ptr(shs), int(sh_degree) if shs is not None else 0,
                    coeffs, C.cast(campos, C.c_void_p), nv.ptr(mask), self.n, capacity,
                    nv.ptr(sc.rec), nv.ptr(sc.rect), nv.ptr(sc.radii), 1 if self.calculate_surface_distance else 0,
                    nv.ptr(sc.cam_key), nv.ptr(sc.cam_surf), nv.ptr(sc.colour), nv.ptr(sc.depths), nv.ptr(sc.invdepths),
                    C.c_void_p(sc.count_host.data_ptr()), int(first), int(step), nv.ptr(sc.back_ws), sc.back_bytes,
                    self._stream_ptr(sc))
            if rc == -4:                               # G2PC_ERR_UNSUPPORTED: a scene the bucket sort does not pay for
                self._fused_ok = fused = False
            else:
                nv.check(rc, "raster_camera_cu")
        if not fused:
            self._front(sc, cam, campos, sh_degree, count_host=False)
            with nv.region("raster_bin+blend_cu", self.device, sc.stream):
                nv.check(L.g2pc_raster_back_cu_dev(
                    C.byref(cam), nv.ptr(mask), self.n, capacity, nv.ptr(sc.rec), nv.ptr(sc.rect), nv.ptr(sc.sorted),
                    nv.ptr(sc.offsets), 1 if self.calculate_surface_distance else 0, nv.ptr(sc.cam_key), nv.ptr(sc.cam_surf),
                    nv.ptr(sc.colour), nv.ptr(sc.depths), nv.ptr(sc.invdepths), C.c_void_p(sc.count_host.data_ptr()),
                    int(first), int(step), nv.ptr(sc.back_ws), sc.back_bytes, self._stream_ptr(sc)), "raster_back_cu_dev")
        if self._last_update is not None and sc.stream is not None:
            sc.stream.wait_event(self._last_update)
        self._back(sc, cam, mask, capacity, cam_index, 4, "raster_update_cu")
        if sc.stream is not None:
            sc.update_done.record(sc.stream)
            self._last_update = sc.update_done

    def _finish(self, entry):
        """Retire a pipelined camera: it is normally long done; one that did not fit its capacity was skipped on the device
        and is rendered again here through the two-call path (exact instance count)."""
        sc, cam, campos, mask, sh_degree, cam_index, capacity = entry
        if sc.stream is not None:
            sc.update_done.synchronize()
        # (u32 words in an int32 tensor: read unsigned -- 0xFFFFFFFF is k_bk_scan's "does not fit 32 bits" sentinel)
        num_rendered, unsorted = int(sc.count_host[0]) & 0xFFFFFFFF, int(sc.count_host[1]) & 0xFFFFFFFF
        self.last = dict(num_rendered=num_rendered)
        if num_rendered > capacity or unsorted != 0:     # too many instances, or depths piled up in one sort bucket
            self.rerendered += 1
            exact = self._render_two_call(sc, cam, campos, mask, sh_degree, cam_index)
            # the capacity grows from the two-call path's EXACT count: after a pile-up the fused call's per-bucket weight sums may
            # have wrapped, and a garbage count would push the capacity past what g2pc_raster_camera_cu accepts
            if exact > capacity:
                self._capacity = max(self._capacity, int(exact * CAPACITY_HEADROOM))
            self.last = dict(num_rendered=exact)

    def _render_two_call(self, sc, cam, campos, mask, sh_degree, cam_index):
        """front -> host reads the instance count -> bin + blend -> ordered update, on the scratch's stream."""
        self._front(sc, cam, campos, sh_degree, count_host=False)
        if sc.stream is not None:
            sc.stream.synchronize()
        num_rendered = int(sc.offsets[self.n].item())
        self._back(sc, cam, mask, num_rendered, cam_index, 3, "raster_bin+blend_cu")
        if self._last_update is not None and sc.stream is not None:
            sc.stream.wait_event(self._last_update)
        self._back(sc, cam, mask, num_rendered, cam_index, 4, "raster_update_cu")
        if sc.stream is not None:
            sc.update_done.record(sc.stream)
            self._last_update = sc.update_done
        return num_rendered

    def flush(self):
        """Complete every camera in flight (pipelined mode) and make the running state visible to the current stream."""
        if not self._pending and not self._pipe:
            return
        while self._pending:
            self._finish(self._pending.pop(0))
        if self.device.type == "cuda" and not nv.emulated():
            cur = torch.cuda.current_stream(self.device)
            for sc in self._pipe:
                cur.wait_stream(sc.stream)

    def __del__(self):
        try:
            self.flush()
        except Exception:
            pass

    def forward(self, raster_settings, return_per_camera=False, cam_index=None, discard_images=False):
        """__init__.py:90-140.  cam_index: global camera order (multi-GPU camera sharding); default = call order.
        discard_images=True (what the point-cloud pipeline wants): nothing is returned and up to PIPELINE_STREAMS
        cameras are kept in flight on separate HIP streams; only the running-state updates are serialised."""
        if cam_index is None:
            cam_index = self._camera_counter
        self._camera_counter = int(cam_index) + 1
        rs = raster_settings
        cam, campos, mask = self._camera(rs)
        self._dirty = True
        n = self.n
        if self.tile_shard is not None:
            # tile split: one camera at a time, with the cross-rank merge between its blend and its state update
            self.flush()
            sc = self._sync
            self._front(sc, cam, campos, rs.sh_degree)
            num_rendered = int(sc.offsets[n].item())
            sc.colour = None
            self._back(sc, cam, mask, num_rendered, cam_index, 3, "raster_bin+blend_cu")
            self._exchange_camera(sc)
            self._back(sc, cam, mask, num_rendered, cam_index, 4, "raster_update_cu")
            if discard_images:
                return None, None, None, None
            return sc.colour, sc.radii.clone(), sc.invdepths, sc.depths
        on_gpu = self.device.type == "cuda" and not nv.emulated()
        if discard_images and PIPELINE_STREAMS > 1 and (on_gpu or PIPELINE_IN_EMULATOR):
            if not self._pipe:
                self._pipe = [_CuScratch(n, self.device, torch.cuda.Stream(self.device) if on_gpu else None, pipelined=True)
                              for _ in range(PIPELINE_STREAMS)]
            while len(self._pending) >= PIPELINE_STREAMS:
                self._finish(self._pending.pop(0))
            sc = self._pipe[self._pipe_next]
            self._pipe_next = (self._pipe_next + 1) % PIPELINE_STREAMS
            if not self._pending and on_gpu:
                for other in self._pipe:
                    other.stream.wait_stream(torch.cuda.current_stream(self.device))
            if mask is not None and on_gpu:
                # the mask was produced (copied / converted) on the CURRENT stream but is read by the blend on sc.stream,
                # possibly several cameras later: order the side stream behind its producer and keep the allocator from
                # handing the block to the next camera's mask while that blend is still queued
                sc.stream.wait_stream(torch.cuda.current_stream(self.device))
                mask.record_stream(sc.stream)
            if self._capacity is None:                 # the first camera tells how many instances to expect
                num_rendered = self._render_two_call(sc, cam, campos, mask, rs.sh_degree, int(cam_index))
                self._capacity = max(int(num_rendered * CAPACITY_HEADROOM), MIN_CAPACITY)
                return None, None, None, None
            self._issue(sc, cam, campos, mask, rs.sh_degree, int(cam_index), self._capacity)
            self._pending.append((sc, cam, campos, mask, rs.sh_degree, int(cam_index), self._capacity))
            return None, None, None, None

        self.flush()
        sc = self._sync
        self._front(sc, cam, campos, rs.sh_degree)
        num_rendered = int(sc.offsets[n].item())              # rasterizer_impl.cu:289 has the same read-back
        cur = (torch.empty(n, dtype=torch.float32, device=self.device), torch.empty(n, dtype=torch.int32, device=self.device),
               torch.empty(n, dtype=torch.float32, device=self.device)) if return_per_camera else (None, None, None)
        sc.colour = None                                      # fresh output tensors: they are handed to the caller
        self._back(sc, cam, mask, num_rendered, cam_index, 7, "raster_back_cu", cur)
        self.last = dict(num_rendered=num_rendered, contributions=cur[0], pixels=cur[1], surface_distances=cur[2])
        return sc.colour, sc.radii.clone(), sc.invdepths, sc.depths

    # the reference's renderer objects are called like functions by the pipeline (gauss_to_pc.py:454)
    def __call__(self, raster_settings, **kwargs):
        if kwargs.pop("return_image", True) is False:
            kwargs["discard_images"] = True
        slot = kwargs.pop("slot", None)
        if slot is not None:
            kwargs["cam_index"] = slot - 1
        return self.forward(raster_settings, **kwargs)

    # ---- getters (__init__.py:160-219) ----------------------------------------------------------------------
    def get_gaussian_colours(self):
        self.flush()
        return self.gaussian_colours * 255

    def get_max_gaussian_contributions(self):
        self.flush()
        return self.gaussian_max_contribution

    def get_total_gaussian_contributions(self):
        self.flush()
        return self.gaussian_total_contribution

    def get_gaussians_above_contribution_threshold(self, contribution_threshold):
        return self.get_max_gaussian_contributions() > contribution_threshold

    def get_gaussians_above_total_contribution_threshold(self, contribution_threshold):
        return self.get_total_gaussian_contributions() > contribution_threshold

    def get_surface_gaussians_below_distance_threshold(self, surface_distance_threshold):
        if not self.calculate_surface_distance:
            raise Exception("Cannot determine Gaussian surface distance as this feature was not set at the start of rendering")
        self.flush()
        surface_indices = (self.gaussian_min_surface_distance < FLT_MAX)
        mean_and_std = torch.std_mean(self.gaussian_min_surface_distance[surface_indices])
        return self.gaussian_min_surface_distance < mean_and_std[1] * surface_distance_threshold

    def get_visible_gaussians(self):
        return self.get_gaussians_above_contribution_threshold(self.visible_gaussian_threshold)

    def get_gaussians_with_low_surface_distance(self):
        return self.get_surface_gaussians_below_distance_threshold(self.surface_distance_std)

    def get_predicted_surface_gaussians(self, predicted_surface_std=0.5):
        return self.get_surface_gaussians_below_distance_threshold(predicted_surface_std)

    def markVisible(self, positions, raster_settings):
        """Frustum test of the native rasteriser (commented out in the reference's binding, __init__.py:79-88; the
        native entry `_C.mark_visible` is still exported there): bool[P], centre in front of the near plane."""
        positions = positions.to(torch.float32).contiguous()
        out = torch.empty((positions.shape[0],), dtype=torch.uint8, device=positions.device)
        view = (C.c_float * 16)(*raster_settings.viewmatrix.reshape(-1).tolist())
        nv.check(nv.lib().g2pc_mark_visible(nv.ptr(positions), positions.shape[0], C.byref(view), nv.ptr(out),
                                            nv.stream_handle(positions.device)), "mark_visible")
        return out.bool()

    def all_reduce_visibility(self, group=None):
        """Multi-GPU, cameras sharded over ranks.  max / min / sum are exact up to the fp32 order of the SUM; the
        winner's colour is selected by (max contribution, then earliest global camera index)."""
        import torch.distributed as dist
        self.flush()
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        if self.tile_shard is not None:       # tile split: the state was merged camera by camera and is already global
            return
        gmax = self.gaussian_max_contribution.clone()
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
        # earliest camera among the ranks that reached the global maximum (strict > in the reference = first wins) ...
        none = torch.full_like(self._winner_cam, 1 << 30)
        cand = torch.where((self.gaussian_max_contribution == gmax) & (gmax > 0), self._winner_cam, none)
        first = cand.clone()
        dist.all_reduce(first, op=dist.ReduceOp.MIN, group=group)
        # ... and ONE rank among those holding it (after an earlier exchange every rank does): the exchange is idempotent
        rank = dist.get_rank(group)
        owner = torch.where((cand == first) & (first < (1 << 30)), torch.full_like(first, rank), none)
        dist.all_reduce(owner, op=dist.ReduceOp.MIN, group=group)
        mine = owner == rank
        self.gaussian_colours = torch.where(mine.unsqueeze(1), self.gaussian_colours, torch.zeros_like(self.gaussian_colours))
        dist.all_reduce(self.gaussian_colours, op=dist.ReduceOp.SUM, group=group)
        # running SUM: only what this rank added since the previous exchange travels; the part every rank already shares
        # (the previous exchange's result) is added back afterwards
        base = getattr(self, "_total_base", None)
        new = self.gaussian_total_contribution - base if base is not None else self.gaussian_total_contribution.clone()
        dist.all_reduce(new, op=dist.ReduceOp.SUM, group=group)
        self.gaussian_total_contribution = new + base if base is not None else new
        self._total_base = self.gaussian_total_contribution.clone()
        dist.all_reduce(self.gaussian_min_surface_distance, op=dist.ReduceOp.MIN, group=group)
        self._winner_cam = first
        self.gaussian_max_contribution = gmax

    def rebase_keys(self):
        pass
