"""
Drop-in mirror of the reference's ``gauss_render.py`` façade: ``get_renderer(...)`` returns an object with
``__call__(camera) -> (render, radii, invdepth, depth)`` and the getters the pipeline uses
(``get_gaussian_colours``, ``get_visible_gaussians``, ``get_total_gaussian_contributions``, ...;
call sites gauss_to_pc.py:429-513).  Rendering runs in libg2pc.so (HIP, gfx950):

  renderer_type "python"        -> the pure-torch renderer's SEMANTICS (gauss_render.py:215-465: quad-tree leaf
                                   tiles pinned to max_tile_size=60 / max_gaussians_per_tile=60000, the defaults of
                                   ``render()``, leaves over the limit split and empty nodes not descended into as
                                   the reference's queue does; strict rect overlap; alpha clip only; colour of the
                                   winning tile) on the tile-binned HIP rasteriser.  This is the parity target of the project.
  renderer_type "cuda" / "hip"  -> the native rasteriser's semantics (16x16 tiles, alpha cut-offs, surface distance).

There is no torch fallback: without libg2pc.so every call raises.
"""
import ctypes as C
from math import tan

import numpy as np
import torch

from g2pc import _native as nv
from g2pc import ops, tiles

# Constant values for calculating spherical harmonics (gauss_render.py:10-41)
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]

# Transmittance floor of the blend (include/g2pc.h, g2pc_raster_back_py).  0.0 = the reference's semantics to the
# letter: every Gaussian of a tile is blended into every pixel, by the kernel that keeps the reference's operation order
# in the exponent (k_blend_py_pk).  With a floor t > 0 (the default) a chunk stops once all its pixels have T < t, a
# (Gaussian, 8x8) visit whose alpha stays below 2^-25 is dropped (T(1 - alpha) == T in fp32 there) and the exponent is
# evaluated in expanded form about the sub-block centre (k_blend_py_dl: <= 2e-5 relative in alpha).  Contributions >= t
# agree with the exact mode to ~1e-6 -- the visibility mask, the culled index set and the point allocation are unchanged
# for thresholds > t (measured at 1 M Gaussians: 0 mask flips) -- and pixel colours move by < t.  One visible difference:
# a Gaussian ALL of whose contributions are below 2^-25 is never seen (key 0, no colour), where the reference's strict
# `>` against the initial 0 marks it as rendered with a contribution < 3e-8; it is far below any usable threshold.
DEFAULT_T_FLOOR = 1e-6
AUTO_SLOT_EPOCH = 63        # camera slots per key epoch when the renderer numbers the cameras itself (= the epoch at 14 tile bits)
BLEND_SUBBLOCKS = None     # 8x8 sub-blocks per blend wave (None -> g2pc.tiles.SUBBLOCKS_PER_CHUNK)
RENDER_STATS = []          # (instances L, tile-sort passes, W*H) of every camera rendered (bench.py reads this)


class _Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("focal_x", C.c_float), ("focal_y", C.c_float), ("width", C.c_int32), ("height", C.c_int32),
                ("bg", C.c_float * 3), ("lim_x", C.c_float), ("lim_y", C.c_float)]


class _Job(C.Structure):
    """G2pcCameraJob: what a captured camera graph reads from device memory."""
    _fields_ = [("cam", _Camera), ("camera_slot", C.c_uint32), ("t_floor", C.c_float), ("tilebuf_lo", C.c_uint32),
                ("tilebuf_hi", C.c_uint32), ("reserved", C.c_uint32), ("alive_lo", C.c_uint32), ("alive_hi", C.c_uint32)]


class _Layout(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("xs", C.c_void_p), ("ws", C.c_void_p), ("ys", C.c_void_p),
                ("hs", C.c_void_p), ("tile_seq", C.c_void_p), ("seq_tile", C.c_void_p), ("tile_pix_off", C.c_void_p),
                ("num_chunks", C.c_int32), ("chunk_tile", C.c_void_p), ("chunk_pix0", C.c_void_p),
                ("chunk_subblocks", C.c_int32), ("seq_bits", C.c_int32),
                ("seq_base", C.c_int32), ("seq_count", C.c_int32), ("tile_mask", C.c_void_p), ("depth", C.c_int32),
                ("inner_x", C.c_void_p), ("inner_y", C.c_void_p), ("tile_stick", C.c_void_p), ("tile_force", C.c_void_p),
                ("tile_parent", C.c_void_p)]


nv._RASTER_PROTOS.update({
    "g2pc_raster_front_workspace": (C.c_size_t, [C.c_int64]),
    "g2pc_raster_front_py": (C.c_int, [C.POINTER(_Camera), C.POINTER(_Layout)] + [C.c_void_p] * 4 + [C.c_int64] +
                             [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]),
    "g2pc_raster_back_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "g2pc_raster_back_py": (C.c_int, [C.POINTER(_Camera), C.POINTER(_Layout), C.c_int64, C.c_int64] +
                            [C.c_void_p] * 6 + [C.c_uint32, C.c_float] + [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "g2pc_raster_tile_states": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "g2pc_raster_node_counts": (C.c_int, [C.POINTER(_Camera), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "g2pc_raster_repack_keys": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "g2pc_raster_camera_workspace": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "g2pc_raster_camera_py": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(_Layout)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] +
                              [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "g2pc_raster_cameras_py": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(_Layout)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] +
                               [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "g2pc_raster_camera_update_py": (C.c_int, [C.POINTER(_Layout), C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "g2pc_graph_capture_begin": (C.c_int, [C.c_void_p]),
    "g2pc_graph_capture_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "g2pc_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "g2pc_graph_destroy": (C.c_int, [C.c_void_p]),
    "g2pc_raster_rebase_keys": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "g2pc_raster_resolve_colours_py": (C.c_int, [C.POINTER(_Layout), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "g2pc_raster_key_owner": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "g2pc_raster_keep_winner_colours": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "g2pc_raster_contributions": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
})
if nv._LIB is not None:
    nv._bind(nv._LIB)


homogeneous = lambda points: torch.cat([points, torch.ones_like(points[..., :1])], dim=-1)     # gauss_render.py:41


def _host16(m):
    """A 4x4 torch / numpy matrix as 16 host floats (row-major), the way the C ABI takes camera matrices."""
    v = m.detach().to("cpu", torch.float32).reshape(-1).tolist() if isinstance(m, torch.Tensor) else np.asarray(m, np.float32).reshape(-1).tolist()
    if len(v) != 16:
        raise ValueError("expected a 4x4 matrix")
    return (C.c_float * 16)(*v)


def eval_sh(deg, sh, dirs=None):
    """gauss_render.py:43-99: evaluate spherical harmonics of degree `deg` (0..4) at unit directions.
    sh [..., C, K >= (deg+1)^2], dirs [..., 3] -> [..., C].  Runs in libg2pc.so (g2pc_eval_sh)."""
    assert deg <= 4 and deg >= 0
    coeff = (deg + 1) ** 2
    assert sh.shape[-1] >= coeff
    if deg > 0:
        assert dirs is not None
    batch = tuple(sh.shape[:-2])
    channels, k = int(sh.shape[-2]), int(sh.shape[-1])
    s = ops._f32c(sh).reshape(-1, channels, k)
    n = s.shape[0]
    d = None
    if dirs is not None:
        d = ops._f32c(dirs.expand(*batch, 3) if tuple(dirs.shape[:-1]) != batch else dirs).reshape(-1, 3)
    out = torch.empty((n, channels), dtype=torch.float32, device=s.device)
    nv.check(nv.lib().g2pc_eval_sh(int(deg), nv.ptr(s), nv.ptr(d), n, channels, k, nv.ptr(out), nv.stream_handle(s.device)),
             "eval_sh")
    return out.reshape(*batch, channels)


def build_covariance_2d(mean3d, cov3d, viewmatrix, fov_x, fov_y, focal_x, focal_y):
    """gauss_render.py:101-148: EWA-splatting 2-D covariances [n,2,2] (low-pass filter 0.3 included)."""
    m, c = ops._f32c(mean3d).reshape(-1, 3), ops._f32c(cov3d).reshape(-1, 3, 3)
    n = m.shape[0]
    out = torch.empty((n, 2, 2), dtype=torch.float32, device=m.device)
    nv.check(nv.lib().g2pc_build_covariance_2d(nv.ptr(m), nv.ptr(c), n, C.cast(_host16(viewmatrix), C.c_void_p),
                                               tan(fov_x * 0.5) * 1.3, tan(fov_y * 0.5) * 1.3, float(focal_x), float(focal_y),
                                               nv.ptr(out), nv.stream_handle(m.device)), "build_covariance_2d")
    return out


def projection_ndc(points, viewmatrix, projmatrix):
    """gauss_render.py:151-168 -> (p_proj [n,4], p_view [n,4], in_mask bool[n])."""
    p = ops._f32c(points).reshape(-1, 3)
    n = p.shape[0]
    p_proj = torch.empty((n, 4), dtype=torch.float32, device=p.device)
    p_view = torch.empty((n, 4), dtype=torch.float32, device=p.device)
    mask = torch.empty((n,), dtype=torch.uint8, device=p.device)
    nv.check(nv.lib().g2pc_projection_ndc(nv.ptr(p), n, C.cast(_host16(viewmatrix), C.c_void_p),
                                          C.cast(_host16(projmatrix), C.c_void_p), nv.ptr(p_proj), nv.ptr(p_view),
                                          nv.ptr(mask), nv.stream_handle(p.device)), "projection_ndc")
    return p_proj, p_view, mask.to(torch.bool)


@torch.no_grad()
def get_radius(cov2d):
    """gauss_render.py:171-180: 2-D radii of the Gaussians."""
    c = ops._f32c(cov2d).reshape(-1, 2, 2)
    out = torch.empty((c.shape[0],), dtype=torch.float32, device=c.device)
    nv.check(nv.lib().g2pc_get_radius(nv.ptr(c), c.shape[0], nv.ptr(out), nv.stream_handle(c.device)), "get_radius")
    return out


@torch.no_grad()
def get_rect(pix_coord, radii, width, height):
    """gauss_render.py:183-193: pixel rectangles (rect_min, rect_max), each [n,2], clipped to the image."""
    p, r = ops._f32c(pix_coord).reshape(-1, 2), ops._f32c(radii).reshape(-1)
    n = p.shape[0]
    rect_min = torch.empty((n, 2), dtype=torch.float32, device=p.device)
    rect_max = torch.empty((n, 2), dtype=torch.float32, device=p.device)
    nv.check(nv.lib().g2pc_get_rect(nv.ptr(p), nv.ptr(r), n, float(width), float(height), nv.ptr(rect_min),
                                    nv.ptr(rect_max), nv.stream_handle(p.device)), "get_rect")
    return rect_min, rect_max


def strip_lowerdiag(L):
    """gauss_render.py:195-205."""
    idx = torch.tensor([0, 1, 2, 4, 5, 8], device=L.device)
    return L.reshape(L.shape[0], 9).index_select(1, idx).to(torch.float)


def strip_symmetric(sym):
    return strip_lowerdiag(sym)


class _DeviceLayout:
    """A tile layout uploaded once per (image size, tiling) and kept alive with its ctypes mirror.  seq_count != 0: a level of a
    camera's data-dependent quad-tree -- its keys carry the sequence numbers seq_base + tile_seq (G2pcTileLayout.seq_base)."""

    def __init__(self, lay, device, seq_base=0, seq_count=0):
        self.host = lay
        self.device = device
        names = ["xs", "ws", "ys", "hs", "tile_seq", "seq_tile", "tile_pix_off", "chunk_tile", "chunk_pix0"]
        self.has_tree = int(lay.get("depth", 0)) > 0
        # leaves one pixel thin (tiles.python_quadtree_layout): the empty-node rule is decided on the HOST for this layout -- the
        # device gate gets no tree (depth 0), every camera takes the two-call path
        self.host_vacancy = self.has_tree and bool(lay.get("thin_leaves"))
        if self.has_tree:
            names += ["inner_x", "inner_y", "tile_stick"]
        self.forced = lay.get("tile_force")            # nodes still larger than max_tile_size: split for every camera
        if self.forced is not None:
            names += ["tile_force"]
        self.t = {k: torch.from_numpy(np.ascontiguousarray(lay[k])).to(device) for k in names}
        if seq_count:
            self.t["tile_seq"] = self.t["tile_seq"] + int(seq_base)
        # tile-sequence bits its keys need: the leaves plus a quarter as many children of split leaves (a camera that needs more
        # widens the renderer's keys in place -- possible only while the camera slots in use fit the wider field's range)
        T = lay["nx"] * lay["ny"]
        reserve = T // 4 if lay.get("tile_force") is None else 4 * int(np.count_nonzero(lay["tile_force"]))   # children to come
        self.seq_bits = min(14, max(12, int(np.ceil(np.log2(max(T + reserve, 2))))))
        self.c = _Layout(nx=lay["nx"], ny=lay["ny"], num_chunks=len(lay["chunk_tile"]),
                         chunk_subblocks=lay["chunk_subblocks"], seq_bits=self.seq_bits, seq_base=int(seq_base),
                         seq_count=int(seq_count), depth=int(lay.get("depth", 0)) if (self.has_tree and not self.host_vacancy) else 0,
                         **{k: v.data_ptr() for k, v in self.t.items()})
        self.num_tiles = lay["nx"] * lay["ny"]
        self.total_pixels = lay["total_pixels"]
        self.sticks_out = self.has_tree and bool(np.any(lay["tile_stick"]))

    def only(self, enabled):
        """The same layout restricted to the tiles with enabled[t]: the blend walks only their chunks and the image
        assembly paints only their pixels (G2pcTileLayout.tile_mask)."""
        return _PassLayout(self, np.asarray(enabled, dtype=bool))

    def child_pass(self, max_tile_size, max_per_tile=0):
        """The child level of this layout as a SECOND captured pass of a camera (instead of host-driven passes at flush()), or None.
          static    image sizes whose size-driven tree is not of uniform depth (tile_force): the nodes still too large are the same
                    for every camera.  Pass A = this layout without the chunks of those nodes, tile_force = 2 ("children follow":
                    the gate empties the node without reporting the camera); pass B = the child level of ALL of them, for every
                    camera.
          on demand (uniform trees, max_per_tile set) the leaves a camera OVERLOADS differ from camera to camera: pass A = this
                    layout as it is (the gate leaves an overloaded leaf out and reports the camera); pass B = the child level of
                    ALL leaves, staged only for the cameras that reported.
        In both, pass A's gate writes "this tile is split" per tile into the camera's `alive` bytes and pass B (tile_parent) exists
        for the children of those tiles only: no instance is emitted for any other tile, and the gate reports (state 1 -> the host
        path at flush()) children that are still too large or hold too many Gaussians.  Sequence numbers: the children of ALL
        parent tiles in the parents' FIFO order continue the leaves' -- order-isomorphic to the reference's queue, which numbers
        the children of the nodes it splits in that order (gauss_render.py:319-335)."""
        static = self.forced is not None
        if not static and not max_per_tile:
            return None
        if not hasattr(self, "_child_pass"):
            self._child_pass = _ChildPass(self, static, max_tile_size)
        cp = self._child_pass
        return cp if cp.usable else None


class _ChildPass:
    def __init__(self, lay, static, max_tile_size):
        self.lay, self.static, self.max_tile_size = lay, static, max_tile_size
        h = lay.host
        self.is_parent = (np.asarray(lay.forced) != 0) if static else np.ones((lay.num_tiles,), bool)
        self.room = lay.num_tiles + 4 * int(self.is_parent.sum())        # sequence numbers: leaves + (at most) four children each
        self.usable = self.room <= (1 << 14)
        self._built = None
        if static:
            self.usable = self.usable and self.build() is not None        # (several runs: left to the host path)

    def parents(self):
        h = self.lay.host
        nx = h["nx"]
        return [(int(h["xs"][t % nx]), int(h["ys"][t // nx]), int(h["ws"][t % nx]), int(h["hs"][t // nx]), (int(h["tile_seq"][t]),))
                for t in np.nonzero(self.is_parent)[0]]

    def runs(self):
        h = self.lay.host
        return _child_levels(int(max(h["xs"] + h["ws"])), int(max(h["ys"] + h["hs"])), self.parents(), self.lay.num_tiles,
                             self.lay.device)

    def build(self):
        """(pass A layout, pass B layout) -- built on first use (an on-demand pass costs nothing until a camera overloads a leaf)."""
        if self._built is None:
            self._built = False
            lay, h = self.lay, self.lay.host
            nx = h["nx"]
            parents = self.parents()
            tile_of = {(p[0], p[1]): int(t) for p, t in zip(parents, np.nonzero(self.is_parent)[0])}
            runs = self.runs()
            if len(runs) == 1:
                level, children, _ = runs[0]
                if self.static:
                    pa = lay.only(~self.is_parent)
                    pa.t["tile_force"] = torch.from_numpy(np.where(self.is_parent, 2, 0).astype(np.uint8)).to(lay.device)
                    pa.c.tile_force = pa.t["tile_force"].data_ptr()
                else:
                    pa = lay
                is_child = np.zeros((level.num_tiles,), bool)
                parent = np.full((level.num_tiles, 4), -1, dtype=np.int32)       # G2PC_TILE_PARENTS entries per tile
                still = np.zeros((level.num_tiles,), dtype=np.uint8)
                # children come four per parent, in the parents' FIFO order (tiles.child_layout): order[:-1] names the parent.
                # Two neighbouring parents can have the SAME rectangle among their children (a child reaches one pixel beyond
                # an odd-sized parent): one tile of the level, up to two parents per axis
                by_order = {p[4]: tile_of[(p[0], p[1])] for p in parents}
                for (t, x0, y0, w, hh, order) in children:
                    is_child[t] = True
                    pt = by_order[tuple(order[:-1])]
                    if pt not in parent[t]:
                        free = np.nonzero(parent[t] < 0)[0]
                        if not len(free):
                            return None                                          # (cannot happen: at most 2 x 2 parents)
                        parent[t, free[0]] = pt
                    still[t] = 1 if (w > self.max_tile_size or hh > self.max_tile_size) else 0
                pb = level.only(is_child)
                pb.t["tile_parent"] = torch.from_numpy(np.ascontiguousarray(parent)).to(lay.device)
                pb.c.tile_parent = pb.t["tile_parent"].data_ptr()
                if still.any():
                    pb.t["tile_force"] = torch.from_numpy(still).to(lay.device)
                    pb.c.tile_force = pb.t["tile_force"].data_ptr()
                pb.children = len(children)
                self.children_of = np.bincount([by_order[tuple(c[5][:-1])] for c in children], minlength=lay.num_tiles)  # per parent tile
                self._built = (pa, pb)
        return self._built or None


class _PassLayout:
    """One pass of a camera whose quad-tree departs from the leaf grid: a layout with some of its tiles switched off."""

    def __init__(self, base, enabled):
        keep = enabled[base.host["chunk_tile"]]
        dev = base.device
        self.base = base                      # (keeps the shared tables alive)
        self.t = dict(chunk_tile=torch.from_numpy(np.ascontiguousarray(base.host["chunk_tile"][keep])).to(dev),
                      chunk_pix0=torch.from_numpy(np.ascontiguousarray(base.host["chunk_pix0"][keep])).to(dev),
                      tile_mask=torch.from_numpy(enabled.astype(np.uint8)).to(dev))
        self.c = _Layout.from_buffer_copy(base.c)
        self.c.num_chunks = int(keep.sum())
        for k, v in self.t.items():
            setattr(self.c, k, v.data_ptr())
        self.num_tiles, self.total_pixels = base.num_tiles, base.total_pixels


class _Scratch:
    """Scratch of the synchronous two-call path (image requested, first camera, emulator)."""

    def __init__(self, n, device):
        f32, i32 = dict(dtype=torch.float32, device=device), dict(dtype=torch.int32, device=device)
        self.rec = torch.empty((n, 16), **f32)                    # one 64-byte blend record per Gaussian
        self.rect, self.sorted_idx = torch.empty((n,), **i32), torch.empty((n,), **i32)
        self.offsets = torch.empty((n + 1,), **i32)
        self.front_ws_bytes = nv.lib().g2pc_raster_front_workspace(n)
        self.front_ws = nv.workspace(self.front_ws_bytes, device)
        self.back_ws, self.back_ws_bytes = None, 0
        self.tilebuf = None
        self.ptrs = tuple(nv.ptr(t) for t in (self.rec, self.rect, self.sorted_idx, self.offsets))
        self.front_ws_ptr = nv.ptr(self.front_ws)


# Cameras in flight when the caller does not need the image back (the pipeline of gauss_to_pc.py discards it).
# Each in-flight camera owns a HIP stream, a device-resident G2pcCameraJob and ONE captured hipGraph holding its
# ~35 launches (preprocess, depth sort, binning, blend): per camera the host rewrites a pinned 200-byte struct and
# issues one graph launch plus the (camera-ordered) colour update -- no read-back, no per-kernel launch cost.
# The packed-key atomicMax makes the blends of different cameras commutative, so camera c+1's small sort / scan
# kernels (which leave most CUs idle) and even its blend overlap camera c's blend.
PIPELINE_STREAMS = 4              # batches in flight (one HIP stream, one captured graph each)
CAMERA_BATCH = 2                  # cameras per launch sequence (g2pc_raster_cameras_py): every kernel runs with grid.y = batch
# How the batches in flight share the device:
#   "chain": every slot owns a stream and replays head + blend as one graph on it.  Slots started together stay in step --
#            all heads (which leave most of the device idle), then all blends (which then share its throughput).
#   "split": ONE head stream (high priority) and ONE blend stream for all slots: the heads of batch i+1 run beside the blends
#            of batch i, blends run back to back; a slot's arena is handed from one stream to the other with events.
#   "split_multi": every slot's heads on its OWN high-priority stream (the small head kernels of several batches run side
#            by side), all blends on ONE shared stream, back to back.
PIPELINE_MODE = "chain"
PIPELINE_SLOTS_PER_STREAM = 1     # batch slots (pinned jobs, arena, captured graphs) per stream: with 2 the next batch of a stream is
                                  # queued while its previous one still runs, so the stream never waits ~0.5 ms for the host to
                                  # notice that a batch is through, stage the next cameras and launch.  Measured (r03zv): no gain
                                  # (17.3-17.6 against 16.9-17.6 ms) -- the kernels of the filled gaps slow the others down
PIPELINE_BLEND_STREAMS = 1        # split modes: blend streams the batches alternate over (2: two blends in flight -- the
                                  # throughput phase of one beside the draining tail of the other)
PIPELINE_IN_EMULATOR = False      # tests: drive the capture / replay path through the CPU emulator too
MAX_GRAPHS_PER_SLOT = 12          # executable graphs a slot keeps (one per layout / pass / batch size met): bounded, ADVICE r04
CAPACITY_HEADROOM = 1.25          # instance capacity of the captured graphs relative to the largest count seen so far
MIN_CAPACITY = 1 << 16
_LAYOUT_CACHE = {}
# Child levels of a camera's data-dependent quad-tree, keyed by (image size, the parents split, first sequence number, device):
# cameras of one job mostly split the SAME nodes (image sizes with a non-uniform size-driven tree split the same interior
# nodes for every camera), and building a level on the host (tiles.child_layout + its chunk lists) costs 3 - 20 ms.
_CHILD_LEVEL_CACHE = {}
_CHILD_LEVEL_CACHE_MAX = 64


def _child_levels(W, H, parents, seq_next, device):
    """[(level _DeviceLayout, children, gate pass-layout)] for the runs of tiles.child_layout (usually one), cached."""
    key = (W, H, BLEND_SUBBLOCKS, int(seq_next), str(device), tuple(sorted((p[4], p[0], p[1], p[2], p[3]) for p in parents)))
    hit = _CHILD_LEVEL_CACHE.get(key)
    if hit is None:
        hit, base = [], int(seq_next)
        for host, children in tiles.child_layout(W, H, parents, BLEND_SUBBLOCKS):
            if host["nx"] > 256 or host["ny"] > 256:
                raise NotImplementedError("a quad-tree level with more than 256 tile intervals per axis")
            level = _DeviceLayout(host, device, seq_base=base, seq_count=len(children))
            is_child = np.zeros((level.num_tiles,), bool)
            is_child[[c[0] for c in children]] = True
            level.parts = {}                        # enabled mask (bytes) -> pass layout, see _render_tree
            hit.append((level, children, level.only(is_child)))
            base += len(children)
        if len(_CHILD_LEVEL_CACHE) >= _CHILD_LEVEL_CACHE_MAX:
            _CHILD_LEVEL_CACHE.pop(next(iter(_CHILD_LEVEL_CACHE)))
        _CHILD_LEVEL_CACHE[key] = hit
    return hit


STREAM_FACTORY = None             # experiments only (bench.py --cu-mask-heads): callable(device, kind) -> torch stream, kind "head" / "blend"


def _new_stream(device, kind, priority=0):
    if STREAM_FACTORY is not None:
        return STREAM_FACTORY(device, kind)
    return torch.cuda.Stream(device, priority=priority)


class _GraphSlot:
    """One in-flight camera of the capture-and-replay pipeline."""

    def __init__(self, device, on_gpu, batch=1, stream=None):
        self.on_gpu = on_gpu
        self.batch = int(batch)
        # (slots may share a stream: PIPELINE_SLOTS_PER_STREAM)
        self.stream = stream if stream is not None else (
            _new_stream(device, "head", priority=-1 if PIPELINE_MODE.startswith("split") else 0) if on_gpu else None)
        self.stream_ptr = C.c_void_p(self.stream.cuda_stream) if on_gpu else C.c_void_p(1)   # the emulator ignores streams
        nbytes = C.sizeof(_Job) * self.batch
        self.job_host = torch.zeros((nbytes,), dtype=torch.uint8)
        self.count_host = torch.zeros((4 * self.batch,), dtype=torch.int32)      # per camera: [instances, depth-bucket-sort overflow, load of an overloaded leaf, -]
        if on_gpu:
            self.job_host, self.count_host = self.job_host.pin_memory(), self.count_host.pin_memory()
        self.jobs = (_Job * self.batch).from_address(self.job_host.data_ptr())
        self.job = self.jobs[0]
        self.job_dev = torch.zeros((nbytes,), dtype=torch.uint8, device=device)
        self.fill = 0                          # cameras written into job_host and not launched yet
        self.fill_lay = None
        self.update_done = torch.cuda.Event() if on_gpu else None
        self.head_done = torch.cuda.Event() if on_gpu else None
        self.graph, self.graph_key = C.c_void_p(None), None
        self.graphs = {}                       # key -> executable graph: a short last batch does not evict the full-batch graph
        self.ws, self.ws_bytes, self.tilebuf = None, 0, None
        self.inflight = None                   # [(camera struct, layout, slot, capacity)] of the batch replay in flight
        self.staged = []                       # the same for the cameras staged in job_host

    def release(self):
        """Destroy every executable graph of the slot (the current one and the cached ones of other batch sizes)."""
        L = nv.lib()
        graphs = [g for g in getattr(self, "graphs", {}).values() if g]
        if self.graph and not any(g.value == self.graph.value for g in graphs):
            graphs.append(self.graph)
        if graphs and self.on_gpu:
            self.stream.synchronize()              # never destroy an executable graph that may still be in flight
        for g in graphs:
            L.g2pc_graph_destroy(g)
        self.graphs = {}
        self.graph, self.graph_key = C.c_void_p(None), None


class _RenderContext:
    """Device-side state of one renderer that outlives it: the scene copies the kernels read, the running state they
    write, the per-stream slots with their workspaces and CAPTURED GRAPHS (which bake in all of those addresses), and
    the instance capacity learned so far.  A process that converts scene after scene of the same size (a service, the
    bench's repeated job) gets the context of the previous renderer back from a small pool instead of paying the
    capture (~0.7 ms), the first camera through the host-synchronised path and the teardown again."""

    def __init__(self, n, device):
        f32 = dict(dtype=torch.float32, device=device)
        self.n, self.device = n, device
        self.means3D, self.cov3d = torch.empty((n, 3), **f32), torch.empty((n, 3, 3), **f32)
        self.opacity, self.colour = torch.empty((n,), **f32), torch.empty((n, 3), **f32)
        self.best_key = torch.empty((n,), dtype=torch.int64, device=device)
        self.gaussian_colours = torch.empty((n, 3), **f32)
        self.overflow = torch.empty((1,), dtype=torch.int32, device=device)
        self.sync_scratch = _Scratch(n, device)
        self.slots, self.capacity = [], None
        self._blend_streams = []
        self.cam_tilebufs = []        # ring of per-tile colour buffers, one per pipelined camera whose colours are still to be
                                      # resolved (deferred colour resolve); bounded by DEFERRED_BUDGET_BYTES, reused by the
                                      # next batch / job of this context

    def blend_stream(self, device, index=0):
        while len(self._blend_streams) <= index:
            self._blend_streams.append(_new_stream(device, "blend"))
        return self._blend_streams[index]

    def release(self):
        for sl in self.slots:
            sl.release()
        self.slots = []
        self.cam_tilebufs = []


# Deferred colour resolve: every pipelined camera keeps a per-tile colour buffer (12 B per pixel) until flush() gives each
# Gaussian the colour of the camera that holds its key.  The set is bounded: once the buffers of the cameras in flight would
# exceed this budget (or DEFERRED_MAX cameras) the renderer resolves what it has and recycles the ring -- 64 cameras at
# 1280x720 (0.7 GB), 20 at 3840x2160, never fewer than 8.
DEFERRED_BUDGET_BYTES = 2 << 30
DEFERRED_MAX = 64
DEFERRED_MIN = 8
CONTEXT_POOL_SIZE = 2
import os as _os
_EXPERIMENT_SKIP = int(_os.environ.get("G2PC_POOL_SKIP_FIRST_JOBS", "0"))     # experiment knob (tools/experiments/first_context.sh)
_JOBS_CLOSED = 0
_CONTEXT_POOL = []            # free contexts, most recently used last


def _acquire_context(n, device):
    for i in range(len(_CONTEXT_POOL) - 1, -1, -1):
        c = _CONTEXT_POOL[i]
        if c.n == n and c.device == device:
            return _CONTEXT_POOL.pop(i)
    clear_context_pool()          # a scene of another size: the pooled contexts (GBs of scene copies, workspaces and
    return _RenderContext(n, device)   # captured graphs) are dead weight -- free them before allocating the new one


def clear_context_pool():
    """Release every pooled device context (scene copies, per-stream workspaces, captured graphs).  The pool only pays
    off for a process that converts scene after scene of the same size; a one-shot conversion calls this after its
    camera loop so that the sampler and the clean-up stages get the memory back."""
    while _CONTEXT_POOL:
        _CONTEXT_POOL.pop().release()


def _return_context(ctx):
    _CONTEXT_POOL.append(ctx)
    while len(_CONTEXT_POOL) > CONTEXT_POOL_SIZE:
        _CONTEXT_POOL.pop(0).release()


class GaussHipRenderer():
    """Stateful per-scene renderer: keeps, for every Gaussian, the largest blend contribution seen in any
    tile of any camera and the pixel colour rendered where it occurred (gauss_render.py:215-264)."""

    needs_camera_epochs = True         # 8-bit camera-order field in the packed keys (multi-GPU: exchange + rebase per epoch)
    MAX_GAUSSIANS_PER_TILE = 60000     # `render()` defaults the reference's __call__ is pinned to in parity runs
    MAX_TILE_SIZE = 60

    def __init__(self, means3D, opacity, colour, cov3d, white_bkgd=True, visible_gaussian_threshold=0.0,
                 semantics="python", t_floor=None, tile_shard=None):
        if semantics != "python":
            raise NotImplementedError("use gaussian_pointcloud_rasterization.GaussianRasterizer for 'cuda' semantics")
        nv.lib()
        self.white_bkgd = white_bkgd
        self.device = means3D.device
        self.semantics = semantics
        self.visible_gaussian_threshold = visible_gaussian_threshold
        # Transmittance floor: visits below 2^-25 / pixels with T below the floor are dropped, which leaves every
        # contribution >= the floor bit-identical.  The visibility test is the reference's strict `>` against the threshold
        # (gauss_render.py:249-252, running max :387): with a threshold AT OR BELOW the floor a Gaussian whose contributions
        # are all tiny is visible (and coloured) in the reference and would be neither here -- so the default floor applies
        # only above it; the API's own default threshold (0.0, gauss_render.py:467-468) therefore takes the to-the-letter blend.
        if t_floor is None:
            t_floor = DEFAULT_T_FLOOR if float(visible_gaussian_threshold) > DEFAULT_T_FLOOR else 0.0
        self.t_floor = float(t_floor)
        # (rank, world): blend only this rank's share of every camera's tiles (multi-GPU jobs with fewer cameras than
        # ranks; the images returned then hold this rank's tiles only)
        self.tile_shard = tuple(tile_shard) if tile_shard is not None and tile_shard[1] > 1 else None
        n = means3D.shape[0]
        self.n = n

        ctx = self.ctx = _acquire_context(n, self.device)
        self.means3D, self.opacity, self.cov3d, self.colour = ctx.means3D, ctx.opacity, ctx.cov3d, ctx.colour
        self.means3D.copy_(means3D.reshape(n, 3))                        # dtype conversion included
        self.opacity.copy_(opacity.reshape(n))
        self.cov3d.copy_(cov3d.reshape(n, 3, 3))
        self.colour.copy_(colour.reshape(n, 3))

        # running state: packed (contribution bits << 32 | ~order) keys and the colours of the winners
        self.best_key, self.gaussian_colours, self.overflow = ctx.best_key, ctx.gaussian_colours, ctx.overflow
        self.best_key.zero_()
        self.gaussian_colours.zero_()
        self.overflow.zero_()
        self.camera_slot = 0
        # Width of the tile-sequence field of the packed keys (include/g2pc.h, G2pcTileLayout.seq_bits): 12 bits (4096 leaf
        # tiles, 255 cameras per key epoch) until a camera needs more (up to 14: 16384 leaves = 7680 x 4320 at 60-pixel
        # leaves, 63 cameras per epoch).  All keys of the running state share one width: widening rebases them first.
        self.seq_bits = 12

        self.sync_scratch = ctx.sync_scratch
        self.overflow_ptr = nv.ptr(self.overflow)
        self.scene_ptrs = (nv.ptr(self.means3D), nv.ptr(self.cov3d), nv.ptr(self.opacity))
        self.colour_ptr = nv.ptr(self.colour)
        self.slots = ctx.slots        # _GraphSlot per in-flight camera (created lazily, kept with the context)
        self.slot_next = 0
        self.capacity = ctx.capacity  # instance capacity of the captured graphs (learned from the first camera rendered)
        self.deferred = {}            # camera slot -> layout of the pipelined cameras whose colours are resolved at flush()
        self.redo = []                # cameras that did not fit their graph's capacity: (camera struct, layout, slot)
        self.fixups = []              # cameras whose overloaded leaves still need their children rendered: (camera struct, layout, slot)
        self.alive_rows = 0           # rows of the context's `alive` pool handed out since the last flush()
        self.on_demand = {}           # camera slot -> (child pass, alive bytes) of the cameras in flight whose pass B is staged on demand
        self.pass_b = []              # static child passes waiting for flush(): (camera struct, pass-B layout, slot, own layout, alive bytes)
        self.split_leaves = 0         # children of overloaded leaves rendered so far (the reference's count-driven split)
        self.host_driven = 0          # pipelined cameras whose quad-tree levels were walked by the host at flush()
        self.child_pass_cameras = 0   # pipelined cameras whose overloaded leaves' children went through an on-demand child pass
        self.passed_b = []
        self.rerendered = 0           # cameras that overflowed their graph's capacity and went through the two-call path
        self.layouts = {}
        self.last_stats = []          # (instances L, tile-sort passes, W*H) per rendered camera
        self._dirty = False           # rendered since the last all_reduce_visibility
        self._contrib = None          # contributions unpacked from best_key since the last render / exchange

    def state_ptrs(self):
        return nv.ptr(self.best_key), nv.ptr(self.gaussian_colours)

    @staticmethod
    def reference_limits(free_bytes):
        """(max_tile_size, max_gaussians_per_tile) as the reference's __call__ derives them from the free device memory
        (gauss_render.py:440-444: 175 000 bytes per Gaussian, tile side = a thousandth of the count) -- before its retry loop,
        which lowers both by 5 / 5 000 for every allocation that fails (:447-463) and so depends on the GPU it runs on.  This
        renderer follows render()'s defaults (MAX_TILE_SIZE = 60, MAX_GAUSSIANS_PER_TILE = 60 000, what the fixtures are pinned
        to); to follow a particular card instead:  r.MAX_TILE_SIZE, r.MAX_GAUSSIANS_PER_TILE = r.reference_limits(24 << 30)."""
        max_gaussians = int(free_bytes / 175000)
        return max_gaussians // 1000, max_gaussians

    def close(self):
        """Finish the cameras in flight and hand the device-side context back to the pool (idempotent)."""
        ctx = getattr(self, "ctx", None)
        if ctx is None:
            return
        try:
            self.flush()                           # (needs self.ctx: the ring of colour buffers, the blend stream)
            self.ctx = None
            if self.device.type == "cuda" and not nv.emulated():
                torch.cuda.current_stream(self.device).synchronize()     # nothing of this renderer is still running
            for sl in self.slots:
                sl.inflight, sl.staged, sl.fill, sl.fill_lay = None, [], 0, None
            ctx.slots, ctx.capacity = self.slots, self.capacity
            global _JOBS_CLOSED
            _JOBS_CLOSED += 1
            if _JOBS_CLOSED <= _EXPERIMENT_SKIP:
                ctx.release()
            else:
                _return_context(ctx)
        except Exception:
            self.ctx = None
            ctx.release()
            raise

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (gauss_render.py:237-264) -----------------------------------------------------------------
    @property
    def gaussian_max_contribution(self):
        self.flush()
        # (the getters of one job -- visible mask, total contributions -- read the same state: unpacked once per render)
        return self._contributions().clone()          # callers own what they get (the cache below stays private)

    def _contributions(self):
        """Contributions unpacked from the keys, cached until the next render / exchange / rebase (read-only: shared by the
        getters of one job)."""
        self.flush()
        if self._contrib is None:
            out = torch.empty((self.n,), dtype=torch.float32, device=self.device)
            nv.check(nv.lib().g2pc_raster_contributions(nv.ptr(self.best_key), self.n, nv.ptr(out),
                                                        nv.stream_handle(self.device)), "contributions")
            self._contrib = out
        return self._contrib

    def get_gaussian_colours(self):
        self.flush()
        return self.gaussian_colours * 255

    def get_gaussians_above_contribution_threshold(self, contribution_threshold):
        if self.t_floor > 0.0 and float(contribution_threshold) <= self.t_floor and self.camera_slot > 0:
            import warnings
            warnings.warn("contribution threshold %g is not above the transmittance floor %g the cameras were blended with: "
                          "Gaussians whose contributions all lie below the floor are missing from the mask (construct the "
                          "renderer with visible_gaussian_threshold <= the floor, or t_floor=0, for the reference's result)"
                          % (contribution_threshold, self.t_floor))
        return self._contributions() > contribution_threshold

    def get_visible_gaussians(self):
        return self.get_gaussians_above_contribution_threshold(self.visible_gaussian_threshold)

    def get_surface_gaussians(self):
        c = self._contributions()
        return c > torch.mean(c)

    def get_total_gaussian_contributions(self):
        # the python renderer returns the running MAX here (gauss_render.py:261-264)
        return self.gaussian_max_contribution

    # ---- rendering ---------------------------------------------------------------------------------------------
    def _layout(self, width, height):
        # tile layouts depend only on the image size and tiling: built and uploaded once per process and device
        key = (width, height, self.MAX_TILE_SIZE, BLEND_SUBBLOCKS, self.tile_shard, str(self.device))
        if key not in _LAYOUT_CACHE:
            host = tiles.python_quadtree_layout(width, height, self.MAX_TILE_SIZE, BLEND_SUBBLOCKS, self.tile_shard)
            if host["nx"] * host["ny"] > 16384 or host["nx"] > 256 or host["ny"] > 256:
                # 14-bit tile field of the packed visibility keys / 8-bit tile-interval ranges of the per-Gaussian rects
                raise NotImplementedError("%dx%d at max_tile_size=%d needs %d quad-tree leaves; the python-semantics rasteriser "
                                          "supports at most 16384 (images up to 7680 x 4320): lower --colour_quality"
                                          % (width, height, self.MAX_TILE_SIZE, host["nx"] * host["ny"]))
            _LAYOUT_CACHE[key] = _DeviceLayout(host, self.device)
        return _LAYOUT_CACHE[key]

    def all_reduce_visibility(self, group=None):
        """Multi-GPU (cameras sharded over ranks): combine the running state of all ranks.  all-reduce MAX of the
        packed (contribution, ~order) keys -- exact and order-free --, an all-reduce MIN elects ONE rank among
        those holding the winning key (several do after an earlier exchange), every other rank zeroes its colour and an
        all-reduce SUM delivers the winners' colours: one non-zero term per Gaussian, so calling this twice is harmless."""
        import torch.distributed as dist
        self.flush()
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        # the keys of all ranks must share one tile-field width: a rank whose cameras needed a wider one (more leaves, split
        # leaves) widened on its own -- the others follow before the keys meet
        bits = torch.tensor([self.seq_bits], dtype=torch.int32, device=self.device)
        dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=group)
        # can every rank widen to that?  Decided TOGETHER: a rank that cannot must not raise while the others go on to the next
        # collective (they would wait for it for ever)
        ok = torch.tensor([1 if self._seq_room_ok(int(bits.item())) else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise ValueError("visibility exchange: some rank's camera slots do not fit the %d-bit tile field another rank's "
                             "cameras needed (start every rank at renderer.seq_bits = 14, as convert_gaussians_to_pc does)"
                             % int(bits.item()))
        if int(bits.item()) > self.seq_bits:
            self._ensure_seq_room(1 << int(bits.item()))
        global_key = self.best_key.clone()
        dist.all_reduce(global_key, op=dist.ReduceOp.MAX, group=group)
        rank = dist.get_rank(group)
        owner = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        st = nv.stream_handle(self.device)
        nv.check(nv.lib().g2pc_raster_key_owner(nv.ptr(self.best_key), nv.ptr(global_key), self.n, rank, nv.ptr(owner), st),
                 "key_owner")
        dist.all_reduce(owner, op=dist.ReduceOp.MIN, group=group)          # exactly one rank per Gaussian keeps its colour
        nv.check(nv.lib().g2pc_raster_keep_winner_colours(nv.ptr(owner), self.n, rank, nv.ptr(self.gaussian_colours), st),
                 "keep_winner_colours")
        dist.all_reduce(self.gaussian_colours, op=dist.ReduceOp.SUM, group=group)
        self.best_key.copy_(global_key)            # in place: the captured graphs hold this tensor's address
        self._contrib = None

    @property
    def camera_epoch(self):
        """Cameras per key epoch: the camera-order field holds slots 1 .. camera_epoch."""
        return (1 << (20 - self.seq_bits)) - 1

    def rebase_keys(self):
        """Forget the camera order of the current keys (they become "earliest"), freeing the 8-bit order field."""
        self.flush()
        nv.check(nv.lib().g2pc_raster_rebase_keys(nv.ptr(self.best_key), self.n, nv.stream_handle(self.device)), "rebase")
        self.camera_slot = 0
        self._contrib = None

    def _camera_struct(self, camera, cam=None):
        cam = _Camera() if cam is None else cam
        cam.view[:] = camera.world_view_transform.reshape(-1).tolist()
        cam.proj[:] = camera.projection_matrix.reshape(-1).tolist()
        cam.tan_fovx, cam.tan_fovy = tan(camera.FoVx * 0.5), tan(camera.FoVy * 0.5)
        # the frustum clamp of build_covariance_2d: python floats (double) until they meet the f32 tensor (gauss_render.py:128-129)
        cam.lim_x, cam.lim_y = tan(camera.FoVx * 0.5) * 1.3, tan(camera.FoVy * 0.5) * 1.3
        cam.focal_x, cam.focal_y = camera.focal_x, camera.focal_y
        cam.width, cam.height = int(camera.image_width), int(camera.image_height)
        bgv = 1.0 if self.white_bkgd else 0.0
        cam.bg[:] = [bgv, bgv, bgv]
        return cam

    def _front(self, sc, cam, lay):
        with nv.region("raster_front", self.device):
            nv.check(nv.lib().g2pc_raster_front_py(C.byref(cam), C.byref(lay.c), *self.scene_ptrs, self.colour_ptr,
                                                   self.n, *sc.ptrs, None, sc.front_ws_ptr, sc.front_ws_bytes,
                                                   nv.stream_handle(self.device)),
                     "raster_front_py")

    def _back(self, sc, cam, lay, slot, num_inst, image, phases, name):
        L = nv.lib()
        need = L.g2pc_raster_back_workspace(num_inst, lay.num_tiles)
        if need > sc.back_ws_bytes:
            sc.back_ws_bytes = int(need * 1.25)
            sc.back_ws = nv.workspace(sc.back_ws_bytes, self.device)
            sc.back_ws_ptr = nv.ptr(sc.back_ws)
        if sc.tilebuf is None or sc.tilebuf.numel() < lay.total_pixels * 3:
            sc.tilebuf = torch.empty((lay.total_pixels * 3,), dtype=torch.float32, device=self.device)
            sc.tilebuf_ptr = nv.ptr(sc.tilebuf)
        with nv.region(name, self.device):
            nv.check(L.g2pc_raster_back_py(C.byref(cam), C.byref(lay.c), self.n, num_inst, *sc.ptrs,
                                           self.scene_ptrs[0], self.scene_ptrs[1], slot, self.t_floor, self.state_ptrs()[0], self.state_ptrs()[1], sc.tilebuf_ptr,
                                           nv.ptr(image), phases, self.MAX_GAUSSIANS_PER_TILE, self.overflow_ptr, sc.back_ws_ptr,
                                           sc.back_ws_bytes, nv.stream_handle(self.device)),
                     "raster_back_py")

    def _note(self, lay, num_inst, W, H):
        bits = max(1, int(np.ceil(np.log2(max(lay.num_tiles, 2)))))
        self.last_stats.append((num_inst, (bits + 7) // 8, W * H))
        RENDER_STATS.append((num_inst, (bits + 7) // 8, W * H))

    def _render_sync(self, cam, lay, slot, return_image, static_done=False):
        """Two-call path on the current stream: the host reads the instance count between the halves -- and what the gate
        decided for the leaves (g2pc_raster_tile_states): a camera whose quad-tree departs from the leaf grid continues in
        _render_tree.  static_done: the pipeline already blended this camera's leaves (minus the gated ones)."""
        sc = self.sync_scratch
        self._front(sc, cam, lay)
        num_inst = int(sc.offsets[self.n].item())                       # the one read-back per camera
        image = torch.empty((cam.height, cam.width, 3), dtype=torch.float32, device=self.device) if return_image else None
        self._back(sc, cam, lay, slot, num_inst, image, 1, "raster_bin")
        counts, states = self._tile_states(sc, lay, num_inst)
        plan = None
        if states.any() or ((return_image or getattr(lay, "host_vacancy", False)) and lay.has_tree and not counts.all()):
            plan = self._static_plan(cam, lay, counts, states)
        if plan is not None and (plan["overloaded"].any() or plan["dead"].any()):
            self._render_tree(sc, cam, lay, slot, num_inst, image, plan, static_done)
        elif not static_done:
            self._back(sc, cam, lay, slot, num_inst, image, 2, "raster_blend")
            self._back(sc, cam, lay, slot, num_inst, image, 4, "raster_update")
        self._note(lay, num_inst, cam.width, cam.height)
        return image, num_inst

    # ---- the reference's data-dependent quad-tree (gauss_render.py:290-335) beyond the fixed leaf grid ---------------------
    def _tile_states(self, sc, lay, num_inst):
        """(Gaussians per tile, gate state per tile) of the camera just binned in the scratch, as numpy arrays."""
        T = lay.num_tiles
        out = torch.empty((2, T), dtype=torch.int32, device=self.device)
        nv.check(nv.lib().g2pc_raster_tile_states(sc.back_ws_ptr, sc.back_ws_bytes, num_inst, T, nv.ptr(out[0]), nv.ptr(out[1]),
                                                  nv.stream_handle(self.device)), "raster_tile_states")
        host = out.cpu().numpy().astype(np.int64)
        return host[0], host[1]

    def _node_counts(self, cam, nodes):
        """Gaussians per pixel rectangle (x0, y0, w, h) by the reference's membership test (gauss_render.py:306-309)."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int32).reshape(-1, 4)
        dev_nodes = torch.from_numpy(nodes).to(self.device)
        out = torch.empty((len(nodes),), dtype=torch.int32, device=self.device)
        nv.check(nv.lib().g2pc_raster_node_counts(C.byref(cam), self.scene_ptrs[0], self.scene_ptrs[1], self.n, nv.ptr(dev_nodes),
                                                  len(nodes), nv.ptr(out), nv.stream_handle(self.device)), "raster_node_counts")
        return out.cpu().numpy().astype(np.int64)

    def _static_plan(self, cam, lay, counts, states):
        """Which leaves of the fixed grid the reference's queue really reaches for this camera: a leaf over
        max_gaussians_per_tile is split (:319), a node without any Gaussian is painted with the background and its subtree
        never visited (:311-314) -- `dead` leaves, `fills` = the painted rectangles.  A node is occupied if a leaf inside it
        has a member, empty if no leaf below it has one; in between (only leaves reaching beyond it have members, odd splits)
        the device counts its members with the reference's test."""
        h = lay.host
        nx, ny, T = h["nx"], h["ny"], lay.num_tiles
        over = (counts > self.MAX_GAUSSIANS_PER_TILE) if self.MAX_GAUSSIANS_PER_TILE else np.zeros((T,), bool)
        if lay.forced is not None:                 # nodes the size rule still splits (tiles.python_quadtree_layout)
            over = over | ((lay.forced != 0) & (counts > 0))
        dead, fills = np.zeros((T,), bool), []
        d = int(h.get("depth", 0)) if lay.has_tree else 0
        if d > 0:
            Cn = counts.reshape(ny, nx)
            stick = h["tile_stick"].reshape(ny, nx)
            occupied, unsure = [], []
            for k in range(d):
                m, b = 1 << k, 1 << (d - k)
                blk = Cn.reshape(m, b, m, b)
                inside = (((stick >> k) & 1) == 0).reshape(m, b, m, b)
                occ = (blk * inside).sum(axis=(1, 3)) > 0
                for ay, ax in zip(*np.nonzero(~occ & (blk.sum(axis=(1, 3)) > 0))):
                    unsure.append((k, ay, ax))
                occupied.append(occ)
            if unsure:
                ix, iy = h["inner_x"], h["inner_y"]
                nodes = [(ix[(1 << k) - 1 + ax, 0], iy[(1 << k) - 1 + ay, 0], ix[(1 << k) - 1 + ax, 1] - ix[(1 << k) - 1 + ax, 0] + 1,
                          iy[(1 << k) - 1 + ay, 1] - iy[(1 << k) - 1 + ay, 0] + 1) for (k, ay, ax) in unsure]
                for (k, ay, ax), c in zip(unsure, self._node_counts(cam, nodes)):
                    occupied[k][ay, ax] = c > 0
            alive = np.ones((1, 1), bool)
            for k in range(d):
                ix, iy = h["inner_x"], h["inner_y"]
                for ay, ax in zip(*np.nonzero(alive & ~occupied[k])):
                    fills.append((ix[(1 << k) - 1 + ax, 0], iy[(1 << k) - 1 + ay, 0], ix[(1 << k) - 1 + ax, 1], iy[(1 << k) - 1 + ay, 1]))
                alive = (alive & occupied[k]).repeat(2, axis=0).repeat(2, axis=1)
            dead = ~alive.reshape(-1)
            over = over & ~dead
        # the gate (k_tile_gate) took the same decisions for every leaf with members, by other means
        has = counts > 0
        if getattr(lay, "host_vacancy", False):
            dev_dead = dead                        # (the device took no such decision: depth 0 in its copy of the layout)
        else:
            dev_dead = (states & 0xFF) == 2
        if not (np.array_equal(states == 1, over) and np.array_equal(dev_dead & has, dead & has)):
            raise RuntimeError("quad-tree plan: the device gate and the host disagree on %d leaves"
                               % int(((states == 1) != over).sum() + ((((states & 0xFF) == 2) & has) != (dead & has)).sum()))
        return dict(overloaded=over, dead=dead, fills=fills)

    def _seq_room_ok(self, need):
        return need <= 14 and self.camera_slot <= (1 << (20 - need)) - 1

    def _ensure_seq_room(self, top):
        """The packed keys' tile field must hold sequence numbers below `top` (leaves + the children of split leaves)."""
        need = max(12, int(np.ceil(np.log2(max(top, 2)))))
        if need <= self.seq_bits:
            return
        if not self._seq_room_ok(need):
            # a format limit (16 384 leaves + children per camera), or a CALLER-assigned slot beyond the wider layout's epoch
            # (slots the renderer assigns itself never are: AUTO_SLOT_EPOCH); multi-rank jobs reach this decision together
            # (all_reduce_visibility) so that no rank raises while the others enter a collective
            raise ValueError("a camera's quad-tree needs %d leaf sequence numbers: beyond the %s of the packed "
                             "visibility keys (set renderer.seq_bits = 14 before the first camera and keep caller-assigned "
                             "slots within its 63-camera epoch)"
                             % (top, "14-bit tile field" if need > 14 else "camera slots left at that width"))
        nv.check(nv.lib().g2pc_raster_repack_keys(self.state_ptrs()[0], self.n, self.seq_bits, need,
                                                  nv.stream_handle(self.device)), "raster_repack_keys")
        self.seq_bits = need

    def _render_tree(self, sc, cam, lay, slot, num_inst, image, plan, static_done):
        """A camera whose quad-tree departs from the leaf grid, pass by pass in the reference's FIFO order: the background of
        the empty nodes, the leaves the queue reaches (minus the overloaded ones), then level by level the children of the
        split nodes -- each level one layout (tiles.child_layout) whose keys continue the camera's sequence numbers."""
        W, H = cam.width, cam.height
        h = lay.host
        if image is not None:
            image.fill_(1.0)                                            # torch.ones (:287); the image is returned flipped (:402)
            for (x0, y0, x1, y1) in plan["fills"]:
                image[y0:y1 + 1, W - 1 - x1:W - x0, :] = float(cam.bg[0])
        if not static_done:
            leaves = lay.only(~plan["overloaded"] & ~plan["dead"])
            leaves.c.seq_bits = self.seq_bits
            self._back(sc, cam, leaves, slot, num_inst, image, 2, "raster_blend")
            self._back(sc, cam, leaves, slot, num_inst, image, 4, "raster_update")
        nx = h["nx"]
        parents = [(int(h["xs"][t % nx]), int(h["ys"][t // nx]), int(h["ws"][t % nx]), int(h["hs"][t // nx]), (int(h["tile_seq"][t]),))
                   for t in np.nonzero(plan["overloaded"])[0]]
        seq_next = lay.num_tiles
        # A camera that already went through a captured CHILD PASS (_DeviceLayout.child_pass; static_done is that pass) carries
        # keys numbered as the pass numbers them: the children of ALL its parent tiles in the parents' FIFO order, whether a
        # parent was split for this camera or not.  Its first level here is the same level object (same numbers; the children
        # of the tiles this camera did not split are left out); in the static form the children of leaves split for their COUNT
        # follow as a run of their own.
        first, dead_parent, runs_f = None, None, ()
        cp = static_done if isinstance(static_done, _ChildPass) else None
        if cp is not None:
            node = lambda t: (int(h["xs"][t % nx]), int(h["ys"][t // nx]), int(h["ws"][t % nx]), int(h["hs"][t // nx]), (int(h["tile_seq"][t]),))
            runs_f = cp.runs()
            dead_parent = {(int(h["tile_seq"][t]),) for t in np.nonzero(cp.is_parent & ~plan["overloaded"])[0]}   # tiles not split
            counted = [node(t) for t in np.nonzero(plan["overloaded"] & ~cp.is_parent)[0]]
            first = list(runs_f) + (list(_child_levels(W, H, counted, seq_next + len(runs_f[0][1]), self.device)) if counted else [])
        while parents:
            runs = first if first is not None else _child_levels(W, H, parents, seq_next, self.device)   # usually ONE run (tiles.child_layout)
            skip_dead = dead_parent if first is not None else None
            counted_by_flush = len(runs_f) if first is not None else 0   # (a child pass's own level: flush() counted its children)
            first = None
            parents = []
            for ri, (level, children, gate) in enumerate(runs):         # (no run: every child is narrower than 2 pixels, :301)
                self._ensure_seq_room(seq_next + len(children))
                level.c.seq_bits = self.seq_bits
                self._front(sc, cam, level)
                n_inst = int(sc.offsets[self.n].item())
                self._level_inst_max = max(getattr(self, "_level_inst_max", 0), n_inst)   # (capacity of a static child pass)
                # the level's layout is the PRODUCT of the children's column and row intervals: the gate only looks at the tiles
                # that ARE children (tile_mask) -- a non-tree tile over the limit is no "overloaded leaf" and reports no load
                gate.c.seq_bits = self.seq_bits
                self._back(sc, cam, gate, slot, n_inst, image, 1, "raster_bin")
                counts, states = self._tile_states(sc, level, n_inst)
                enabled = np.zeros((level.num_tiles,), bool)
                for (t, x0, y0, w, h_, order) in children:
                    if skip_dead and tuple(order[:-1]) in skip_dead:
                        continue                                        # child of a node that held no Gaussian: never visited
                    too_many = self.MAX_GAUSSIANS_PER_TILE and counts[t] > self.MAX_GAUSSIANS_PER_TILE
                    too_large = counts[t] > 0 and (w > self.MAX_TILE_SIZE or h_ > self.MAX_TILE_SIZE)  # (:319, after the empty test :311)
                    if too_many or too_large:
                        parents.append((x0, y0, w, h_, order))          # split again at the next level
                    else:
                        enabled[t] = True                               # blended (an empty child paints the background, :311-314)
                if self.tile_shard is not None:                         # this rank's share of the children
                    mine = np.zeros_like(enabled)
                    mine[[c[0] for c in children][self.tile_shard[0]::self.tile_shard[1]]] = True
                    enabled &= mine
                if enabled.any():
                    ek = enabled.tobytes()
                    part = level.parts.get(ek)
                    if part is None:
                        if len(level.parts) >= 8:
                            level.parts.clear()
                        part = level.parts[ek] = level.only(enabled)
                    part.c.seq_bits = self.seq_bits
                    self._back(sc, cam, part, slot, n_inst, image, 2, "raster_blend")
                    self._back(sc, cam, part, slot, n_inst, image, 4, "raster_update")
                seq_next += len(children)
                if ri >= counted_by_flush:
                    self.split_leaves += len(children)

    # ---- capture-and-replay pipeline ------------------------------------------------------------------------------
    def _capture(self, sl, lay, key):
        """(Re)build the slot's hipGraph for (layout, capacity, phases, cameras in the batch): buffers first, then one
        recorded batch call."""
        L = nv.lib()
        capacity, batch = key[1], key[3]
        # graphs of the same (layout, capacity) but another number of cameras share the slot's buffers and stay cached;
        # anything else (new capacity, new layout, new buffers) starts the slot afresh
        if any(k[1] != key[1] for k in sl.graphs):
            sl.release()                           # (graphs of ANOTHER layout at the same capacity stay: a static child pass
                                                   # alternates with its cameras' own layout, the buffers only ever grow)
        need = L.g2pc_raster_camera_workspace(self.n, capacity, lay.num_tiles) * sl.batch
        import contextlib
        with (torch.cuda.stream(sl.stream) if sl.on_gpu else contextlib.nullcontext()):   # allocate on the stream using them
            if need > sl.ws_bytes:
                sl.release()                       # the cached graphs hold the old workspace's addresses
                sl.ws = None
                sl.ws_bytes = int(need)
                sl.ws = nv.workspace(sl.ws_bytes, self.device)
            if sl.tilebuf is None or sl.tilebuf.numel() < lay.total_pixels * 3:
                sl.release()                       # ... and the slot's fallback colour buffer's
                sl.tilebuf = torch.empty((lay.total_pixels * 3,), dtype=torch.float32, device=self.device)
        if len(sl.graphs) >= MAX_GRAPHS_PER_SLOT:
            sl.release()                           # a job over many image sizes / passes: start the slot's cache afresh
        # run the state-free half once outside the capture: kernels that are launched for the first time INSIDE a stream
        # capture (k_preprocess_py<true>, k_resolve_count, ...) leave a graph that replays ~25 % slower for good
        nv.check(self._camera_call(sl, lay, capacity, 1, batch), "raster_cameras_py (warm-up)")
        nv.check(L.g2pc_graph_capture_begin(sl.stream_ptr), "graph_capture_begin")
        rc = self._camera_call(sl, lay, capacity, key[2], batch)
        graph = C.c_void_p(None)
        rc_end = L.g2pc_graph_capture_end(sl.stream_ptr, C.byref(graph))
        nv.check(rc or rc_end, "raster_cameras_py (capture)")
        sl.graphs[key] = graph
        sl.graph, sl.graph_key = graph, key

    def _camera_call(self, sl, lay, capacity, phases, batch, stream_ptr=None):
        return nv.lib().g2pc_raster_cameras_py(nv.ptr(sl.job_dev), C.c_void_p(sl.job_host.data_ptr()), int(batch), C.byref(lay.c),
                                               *self.scene_ptrs, self.colour_ptr, self.n, capacity, self.state_ptrs()[0],
                                               nv.ptr(sl.tilebuf), C.c_void_p(sl.count_host.data_ptr()),
                                               self.MAX_GAUSSIANS_PER_TILE, self.overflow_ptr, phases, nv.ptr(sl.ws),
                                               sl.ws_bytes, stream_ptr if stream_ptr is not None else sl.stream_ptr)

    def _retire(self, sl):
        """The slot's previous batch: wait for it (normally long done), collect its counts, queue a re-render of every
        camera that did not fit."""
        if sl.inflight is None:
            return
        batch, sl.inflight = sl.inflight, None
        if sl.on_gpu:
            sl.update_done.synchronize()
        for i, (cam, lay, slot, capacity, orig, second) in enumerate(batch):
            # (the pinned counts are u32 words in an int32 tensor: read them unsigned -- a count >= 2^31, or k_bk_scan's
            # 0xFFFFFFFF "the instance count does not fit 32 bits" sentinel, must not read as negative and pass for a fit)
            num_inst, unsorted, overloaded = (int(sl.count_host[4 * i + j]) & 0xFFFFFFFF for j in range(3))
            if num_inst > capacity or unsorted:
                # did not fit the graph's buffers, or the depth bucket sort met a pile-up of equal depths: the graph skipped
                # the camera as a whole; render it again through the two-call path (radix depth sort, exact instance count)
                if num_inst > capacity and not unsorted and num_inst != 0xFFFFFFFF:
                    # (with a pile-up the per-bucket weight sums may have wrapped: the count is not one to size buffers from)
                    self.capacity = max(self.capacity, int(num_inst * CAPACITY_HEADROOM))
                self.redo.append((cam, orig, slot))            # (the camera's OWN layout: leaves and children, idempotent)
                # a pass A that did not fit takes its pass B with it: the two-call path numbers the children as IT meets them
                # (a pass B that did not fit wrote nothing: the graph skips the camera as a whole)
                self.pass_b = [p for p in self.pass_b if p[2] != slot]
                self.on_demand.pop(slot, None)
                self.rerendered += 1
                continue
            if overloaded:
                # some leaf held more than max_gaussians_per_tile Gaussians (or a child of the static pass is still too large):
                # the graph left it out (k_tile_gate); its children are rendered at flush() -- the packed keys make the order of
                # the passes irrelevant, and re-blending what a static child pass already blended changes nothing
                od = None if second else self.on_demand.get(slot)
                if od is not None and od[0].build() is not None:
                    # on-demand child pass: the children of ALL the leaves this camera overloaded go through the pipeline
                    self.pass_b.append((cam, od[0], slot, orig, od[1]))
                elif not any(f[2] == slot and f[1] is orig for f in self.fixups):
                    self.fixups.append((cam, orig, slot))
            if not second:
                self._note(lay, num_inst, cam.width, cam.height)

    def _launch_batch(self, sl):
        """Replay the slot's graph for the cameras staged in its job array (a short last batch gets its own graph)."""
        L = nv.lib()
        batch, lay = sl.fill, sl.fill_lay
        if batch == 0:
            return
        on_gpu = sl.on_gpu
        if on_gpu and not any(o.inflight for o in self.slots):
            for o in self.slots:
                o.stream.wait_stream(torch.cuda.current_stream(self.device))      # scene tensors / state are ready
            if PIPELINE_MODE.startswith("split"):
                for b in range(max(1, int(PIPELINE_BLEND_STREAMS))):
                    self.ctx.blend_stream(self.device, b).wait_stream(torch.cuda.current_stream(self.device))
        # profiling: HIP events around the blend alone -> the graph stops before it and the blend is issued directly
        # (this runtime refuses event-record nodes inside a captured graph)
        exact = 8 if self.t_floor == 0.0 else 0      # to-the-letter mode: the blend kernel with the reference's operation order
        split = on_gpu and PIPELINE_MODE.startswith("split")
        # (the gate's limit is a kernel argument baked into the captured launches: renderers with another limit -- the pool
        # hands slots and graphs from job to job -- capture their own)
        key = (id(lay), self.capacity, (1 if (nv.PROFILE is not None or split) else 3) | exact, batch, self.seq_bits,
               int(self.MAX_GAUSSIANS_PER_TILE or 0))
        if sl.graph_key != key:
            if key in sl.graphs:
                sl.graph, sl.graph_key = sl.graphs[key], key
            else:
                self._capture(sl, lay, key)
        if split:
            # heads of all slots on the slots' common head stream (slot 0's, high priority), blends on the common blend stream
            head = sl.stream if PIPELINE_MODE == "split_multi" else self.slots[0].stream
            blend = self.ctx.blend_stream(self.device, self.slots.index(sl) % max(1, int(PIPELINE_BLEND_STREAMS)))
            head.wait_stream(sl.stream)                    # (a capture's warm-up run on the slot's own stream)
            head.wait_event(sl.update_done)                # this slot's arena: its previous blends are through
            nv.check(L.g2pc_graph_launch(sl.graph, C.c_void_p(head.cuda_stream)), "graph_launch")
            sl.head_done.record(head)
            blend.wait_event(sl.head_done)
            with nv.region("raster_blend", self.device, blend):
                nv.check(self._camera_call(sl, lay, key[1], 2 | exact, batch, C.c_void_p(blend.cuda_stream)), "raster_cameras_py (blend)")
            sl.update_done.record(blend)
        else:
            nv.check(L.g2pc_graph_launch(sl.graph, sl.stream_ptr), "graph_launch")
            if (key[2] & 3) == 1:
                with nv.region("raster_blend", self.device, sl.stream):
                    nv.check(self._camera_call(sl, lay, key[1], 2 | exact, batch), "raster_cameras_py (blend)")
            if on_gpu:
                sl.update_done.record(sl.stream)           # "this batch's blends are done" (the colours are resolved at flush)
        sl.inflight, sl.staged = [(c, l, s_, key[1], o, sec) for (c, l, s_, o, sec) in sl.staged], []
        sl.fill, sl.fill_lay = 0, None
        self.slot_next = (self.slot_next + 1) % len(self.slots)

    def _render_pipelined(self, camera, lay, slot):
        on_gpu = self.device.type == "cuda" and not nv.emulated()
        if self.redo:
            self.flush()
        if self.capacity is None:                  # the first camera tells how many instances to expect
            self.flush()
            self._level_inst_max = 0
            _, num_inst = self._render_sync(self._camera_struct(camera), lay, slot, False)
            # (an image size with a static child pass: the children's level holds more instances than the leaves')
            self.capacity = max(int(max(num_inst, self._level_inst_max) * CAPACITY_HEADROOM), MIN_CAPACITY)
            return
        # (the batched scans take at most 2 M values per camera; larger scenes keep one camera per launch sequence)
        batch = max(1, min(int(CAMERA_BATCH), 8)) if self.n <= (2 << 20) else 1
        per_stream = max(1, int(PIPELINE_SLOTS_PER_STREAM)) if PIPELINE_MODE == "chain" else 1
        if len(self.slots) != PIPELINE_STREAMS * per_stream or (self.slots and self.slots[0].batch != batch):
            self.flush()
            for sl in self.slots:
                sl.release()
            # slot i and slot i + PIPELINE_STREAMS share stream i: the cameras go A1 B1 C1 D1 A2 B2 ..., so a stream's second
            # slot is staged and launched while its first is still running
            first = [_GraphSlot(self.device, on_gpu, batch) for _ in range(PIPELINE_STREAMS)]
            self.slots[:] = first + [_GraphSlot(self.device, on_gpu, batch, stream=first[i % PIPELINE_STREAMS].stream)
                                     for i in range(PIPELINE_STREAMS * (per_stream - 1))]
            self.slot_next = 0
        cp = lay.child_pass(self.MAX_TILE_SIZE, self.MAX_GAUSSIANS_PER_TILE) if (self.tile_shard is None and hasattr(lay, "child_pass")) else None
        if cp is not None and cp.room > (1 << self.seq_bits):
            # the keys' tile field must hold the children's sequence numbers BEFORE a camera is in flight with the narrower
            # layout (the widening repacks every key and re-captures the graphs)
            if self._seq_room_ok(max(12, int(np.ceil(np.log2(cp.room))))):
                self.flush()
                self._ensure_seq_room(cp.room)
            else:
                cp = None                          # (caller-assigned slots beyond the wider layout's epoch: the host path)
        if cp is None:
            self._stage(camera, lay, slot, lay)
            return
        # Child pass (_DeviceLayout.child_pass).  Static form -- an image size whose size-driven tree is not of uniform depth:
        # pass A now (the leaves; the gate notes which of the nodes still too large it split), pass B -- their children, the same
        # layout for every camera -- when the cameras are flushed.  On-demand form: pass A is the camera's own layout, pass B is
        # staged at flush() for the cameras whose pass A reported an overloaded leaf (_retire).  The packed-key atomicMax makes
        # the order of the passes irrelevant.
        # One row of a persistent [256, tiles] byte array per pending camera (written by pass A's gate -- every entry --, read by
        # pass B).  _stage takes the row, AFTER every flush it can trigger itself (retiring the slot's previous batch grows
        # pass_b, which can reach the deferred-buffer limit): flush() hands all rows back, and a row taken before it would be
        # given to a later camera whose pass-A gate overwrites the bytes this camera's pass B still has to read (ADVICE r04:
        # jobs longer than the limit whose overloaded leaves vary between cameras lost maxima of up to 0.28).
        if cp.static:
            pa, pb = cp.build()
            pa.c.seq_bits = self.seq_bits
            alive = self._stage(camera, pa, slot, lay, take_alive=True)
            self.pass_b.append((self._camera_struct(camera), cp, slot, lay, alive))
        else:
            alive = self._stage(camera, lay, slot, lay, take_alive=True)
            self.on_demand[slot] = (cp, alive)

    def _take_alive_row(self, lay):
        """The next free row of the context's `alive` pool ([256, tiles] bytes, allocated once per renderer context, so no block
        of the caching allocator changes streams under it).  May flush (all rows in use): call it after any other flush."""
        pool = getattr(self.ctx, "alive_pool", None)
        if pool is None or pool.shape[1] < lay.num_tiles:
            self.flush()                           # (rows of the old pool may be pending)
            if self.device.type == "cuda" and not nv.emulated():
                torch.cuda.synchronize(self.device)
            pool = self.ctx.alive_pool = torch.zeros((256, lay.num_tiles), dtype=torch.uint8, device=self.device)
            if self.device.type == "cuda" and not nv.emulated():
                torch.cuda.synchronize(self.device)
        if self.alive_rows >= pool.shape[0]:
            self.flush()
        alive = pool[self.alive_rows]
        self.alive_rows += 1
        return alive

    def _make_deferred_room(self, lay):
        """The deferred colour buffers are bounded (DEFERRED_*): once the cameras waiting for their colours reach the limit of
        this image size, resolve them (flush) and recycle the ring.  True if it flushed."""
        limit = max(DEFERRED_MIN, min(DEFERRED_MAX, DEFERRED_BUDGET_BYTES // (lay.total_pixels * 12)))
        if len(self.deferred) + len(self.pass_b) < limit:
            return False
        self.flush()                               # resolve the colours of the cameras so far; their buffers are free again
        del self.ctx.cam_tilebufs[limit:]          # a smaller image earlier in the job may have grown the ring past this limit
        return True

    def _stage(self, camera, lay, slot, orig, alive=None, second=False, take_alive=False):
        """Write one camera into the next free job of the pipeline (launching the batch when it is full).  lay: the layout this
        pass blends; orig: the camera's own layout (what a re-render through the two-call path uses); alive: see _DeviceLayout.child_pass
        (take_alive: a fresh row of the pool, taken here behind the last flush this call can trigger; returned);
        second: a child pass staged from inside flush() (never flushes itself)."""
        on_gpu = self.device.type == "cuda" and not nv.emulated()
        sl = self.slots[self.slot_next]
        if sl.fill and sl.fill_lay is not lay:
            self._launch_batch(sl)                  # another image size / pass: the staged cameras go as a short batch
            sl = self.slots[self.slot_next]
        if sl.fill == 0:
            self._retire(sl)                        # the job array is rewritten: its previous batch must be through
        # every pipelined camera renders into its OWN per-tile colour buffer (address in the job, not in the graph): the
        # winners' colours are then resolved in one pass at flush() instead of one update per camera chained in camera
        # order across the streams
        ring = self.ctx.cam_tilebufs
        if not second and self._make_deferred_room(lay):
            sl = self.slots[self.slot_next]
        if take_alive:
            alive = self._take_alive_row(orig)
            sl = self.slots[self.slot_next]        # (it may have flushed: every slot is then retired and empty)
        idx = len(self.deferred)
        if idx >= len(ring) or ring[idx].numel() < lay.total_pixels * 3:
            import contextlib
            with (torch.cuda.stream(sl.stream) if on_gpu else contextlib.nullcontext()):       # first written on this stream
                tb = torch.empty((lay.total_pixels * 3,), dtype=torch.float32, device=self.device)
            if idx >= len(ring):
                ring.append(tb)
            else:
                ring[idx] = tb
        tb = ring[idx]
        job = sl.jobs[sl.fill]
        if isinstance(camera, _Camera):
            C.memmove(C.byref(job.cam), C.byref(camera), C.sizeof(_Camera))
        else:
            self._camera_struct(camera, job.cam)                                    # rewrite the pinned job in place
        job.camera_slot, job.t_floor = slot, self.t_floor
        job.tilebuf_lo, job.tilebuf_hi = tb.data_ptr() & 0xFFFFFFFF, tb.data_ptr() >> 32
        ap = alive.data_ptr() if alive is not None else 0
        job.alive_lo, job.alive_hi = ap & 0xFFFFFFFF, ap >> 32
        self.deferred[(slot, id(lay))] = (lay, tb)
        sl.staged.append((_Camera.from_buffer_copy(job.cam), lay, slot, orig, second))
        sl.fill += 1
        sl.fill_lay = lay
        if sl.fill == sl.batch:
            self._launch_batch(sl)
        return alive

    def flush(self):
        """Complete every camera staged or in flight and make the running state visible to the current stream."""
        if not self.slots:
            return
        for sl in self.slots:
            if sl.fill:
                self._launch_batch(sl)             # a short last batch
        for sl in self.slots:
            self._retire(sl)
        self.passed_b = []                         # (slot, child pass) of the cameras whose pass B went through the pipeline
        if self.pass_b:
            # child passes (static: every camera of an image size with a non-uniform size-driven tree; on demand: the cameras
            # that overloaded a leaf): every camera's pass A is through, its `alive` bytes are written; the children go through
            # the same pipeline, batched like cameras
            pending, self.pass_b = self.pass_b, []
            for (cam, cp, slot, orig, alive) in pending:
                pb = cp.build()[1]
                pb.c.seq_bits = self.seq_bits
                self.passed_b.append((slot, cp))
                self._stage(cam, pb, slot, orig, alive=alive, second=True)
            for sl in self.slots:
                if sl.fill:
                    self._launch_batch(sl)
            for sl in self.slots:
                self._retire(sl)
        else:
            pending = []
        self.on_demand = {}
        self.alive_rows = 0                        # (the alive rows are free again)
        if self.device.type == "cuda" and not nv.emulated():
            cur = torch.cuda.current_stream(self.device)
            for sl in self.slots:
                cur.wait_stream(sl.stream)
            for b in self.ctx._blend_streams:
                cur.wait_stream(b)
        redone = {r[2] for r in self.redo}
        for (cam, cp, slot, orig, alive) in pending:
            if slot not in redone:                 # children rendered: those of the tiles the camera split (its alive bytes)
                self.split_leaves += int(cp.children_of[alive[:cp.lay.num_tiles].cpu().numpy() != 0].sum())
                self.child_pass_cameras += 0 if cp.static else 1
        del pending
        while self.redo:                           # cameras that overflowed their graph: two-call path, original slot
            cam, lay, slot = self.redo.pop(0)
            for k in [k for k in self.deferred if k[0] == slot]:
                self.deferred.pop(k)               # ... which updates the colours it wins at once
            # (everything of the camera, children included, numbered as the two-call path meets them: no second numbering)
            self.fixups = [f for f in self.fixups if f[2] != slot]
            self.passed_b = [q for q in self.passed_b if q[0] != slot]
            lay.c.seq_bits = self.seq_bits         # (layouts are shared between renderers)
            self._render_sync(cam, lay, slot, False)
        numbered = dict(self.passed_b)
        while self.fixups:                         # cameras with overloaded leaves: the children of those leaves, original slot
            cam, lay, slot = self.fixups.pop(0)
            lay.c.seq_bits = self.seq_bits
            self.host_driven += 1
            # (a camera that went through a child pass keeps that pass's sequence numbers: _render_tree)
            self._render_sync(cam, lay, slot, False, static_done=numbered.get(slot, True))
            self.last_stats.pop()                  # (noted when its batch retired)
            RENDER_STATS.pop()
        if self.deferred:
            # deferred colour resolve: one pass per layout over the Gaussians, colour = the winner camera's tile buffer
            by_layout = {}
            for (slot, _), (lay, tb) in self.deferred.items():
                by_layout.setdefault(id(lay), (lay, []))[1].append((slot, tb))
            for lay, slots in by_layout.values():
                lay.c.seq_bits = self.seq_bits
                table = np.zeros((256,), dtype=np.uint64)
                for slot, tb in slots:
                    table[slot] = tb.data_ptr()
                table_dev = torch.from_numpy(table.view(np.int64)).to(self.device)
                nv.check(nv.lib().g2pc_raster_resolve_colours_py(C.byref(lay.c), self.n, self.state_ptrs()[0], nv.ptr(table_dev),
                                                               self.state_ptrs()[1], nv.stream_handle(self.device)),
                         "raster_resolve_colours_py")
            self.deferred = {}

    def check_tile_load(self):
        """Largest number of Gaussians met in one leaf tile above max_gaussians_per_tile (0: no leaf was ever split).  Such
        leaves are split as the reference's queue splits them (gauss_render.py:319-335; _render_tree)."""
        worst = int(self.overflow.item())
        return worst

    def __call__(self, camera, return_image=True, slot=None, **kwargs):
        W, H = int(camera.image_width), int(camera.image_height)
        lay = self._layout(W, H)
        self._dirty = True
        self._contrib = None
        if lay.seq_bits > self.seq_bits:           # more leaf tiles than the keys' tile field holds: widen it (keys rebased)
            if self.camera_slot:
                self.rebase_keys()
            self.seq_bits = lay.seq_bits
        lay.c.seq_bits = self.seq_bits             # (layouts are shared between renderers: every call states its width)
        if slot is not None:                       # caller-assigned global camera order (multi-GPU camera sharding)
            if not (1 <= slot <= self.camera_epoch):
                raise ValueError("camera slot must be in [1, %d]" % self.camera_epoch)
            self.camera_slot = int(slot)
        else:
            # slots the renderer assigns itself stay within the epoch of the WIDEST key layout (14-bit tile field: 63 cameras):
            # a camera whose split leaves need a wider tile field later can then always be widened in place, in the middle of a
            # flush too (_ensure_seq_room), whatever the slot in use.  A rebase costs a flush and one pass over the keys.
            if self.camera_slot >= min(self.camera_epoch, AUTO_SLOT_EPOCH):
                self.rebase_keys()
            self.camera_slot += 1
        slot = self.camera_slot

        on_gpu = self.device.type == "cuda" and not nv.emulated()
        if ((not return_image) and PIPELINE_STREAMS > 1 and (on_gpu or (nv.emulated() and PIPELINE_IN_EMULATOR))
                and not getattr(lay, "host_vacancy", False)):
            self._render_pipelined(camera, lay, slot)
            return None, None, None, None

        self.flush()
        image, _ = self._render_sync(self._camera_struct(camera), lay, slot, return_image)
        return image, None, None, None


# the reference's class name (gauss_render.py:210): same constructor signature, same getters, rendering in HIP
GaussPythonRenderer = GaussHipRenderer


def get_renderer(renderer_type: str, xyz, opacities, colours, covariances, shs=None, visible_gaussian_threshold=0.0,
                 surface_distance_std=None, calculate_surface_distance=False, tile_shard=None):
    """gauss_render.py:467-493.  tile_shard (not in the reference): see GaussHipRenderer."""
    if renderer_type in ("cuda", "hip"):
        from gaussian_pointcloud_rasterization import GaussianRasterizer as GaussianPCRasterizer

        means2D = torch.zeros_like(xyz, dtype=xyz.dtype) + 0

        if shs is None:
            return GaussianPCRasterizer(xyz.to(torch.float), means2D, opacities.type(torch.float),
                                        colors_precomp=colours.to(torch.float), cov3D_precomp=strip_symmetric(covariances).to(torch.float),
                                        visible_gaussian_threshold=visible_gaussian_threshold, surface_distance_std=surface_distance_std,
                                        calculate_surface_distance=calculate_surface_distance, tile_shard=tile_shard)
        else:
            return GaussianPCRasterizer(xyz.to(torch.float), means2D, opacities.type(torch.float),
                                        shs=shs.to(torch.float), cov3D_precomp=strip_symmetric(covariances).to(torch.float),
                                        visible_gaussian_threshold=visible_gaussian_threshold, surface_distance_std=surface_distance_std,
                                        calculate_surface_distance=calculate_surface_distance, tile_shard=tile_shard)

    elif renderer_type == "python":
        return GaussHipRenderer(xyz, opacities, colours, covariances, semantics="python",
                                visible_gaussian_threshold=visible_gaussian_threshold, tile_shard=tile_shard)

    raise Exception(f"Renderer of type {renderer_type} is not supported")
