"""
Point-cloud post-processing behind the reference's ``mesh_handler`` interface (mesh_handler.py:42-119).

``clean_point_cloud`` (mesh_handler.py:89-94, CLI flag ``--clean_pointcloud``) is Open3D's statistical outlier removal
(nb_neighbors = 20, std_ratio = 10).  Here it runs on the GPU: exact k-nearest-neighbour mean distances on a uniform grid
(libg2pc.so, csrc/clean.hip), the cloud statistics and the selection in float64 on the device, the reference's
colour round trip (clamp, truncate, /255, *255, truncate) kept.  No Open3D, no host copy of the cloud.

``generate_mesh`` (Poisson reconstruction + Laplacian smoothing, mesh_handler.py:66-87) is Open3D's own algorithm and out
of scope (SURVEY.md §8f stops "up to, not including, Open3D"): the entry point exists and says so.
"""
import ctypes as C

import torch

from g2pc import _native as nv

NB_NEIGHBORS = 20
TARGET_POINTS_PER_CELL = 8.0       # mean occupancy of the non-empty cells the finest grid is refined towards
MAX_CELLS = 1 << 27                # dense cell table: 4 B per cell
SHELL_BUDGET = 3                   # shells of cells a query may search on one level before a coarser grid takes over
LEVEL_FACTOR = 5.0                 # cell edge ratio between consecutive levels of the cascade
BOX_QUANTILE = 0.002               # the grid covers the [q, 1-q] quantile box of the cloud; the rest lands in border cells

nv._RASTER_PROTOS.update({
    "g2pc_outlier_grid_workspace": (C.c_size_t, [C.c_int64]),
    "g2pc_outlier_grid_build": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_float * 3), C.c_float, C.POINTER(C.c_int32 * 3),
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "g2pc_outlier_knn_mean_distance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float * 3), C.c_float,
                                                 C.POINTER(C.c_int32 * 3), C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                                 C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
})
if nv._LIB is not None:
    nv._bind(nv._LIB)

LAST_STATS = {}                    # diagnostics of the latest knn_mean_distance call (levels, queries per level)


def _dims(extent, cell):
    return [max(1, min(int(e / cell) + 1, 1 << 20)) for e in extent]


def _robust_box(pts):
    """[q, 1-q] quantile box per axis (from a subsample), widened by 2 %: a handful of far-away floaters must not
    dictate the cell size of the grid that serves the bulk of the cloud."""
    m = pts.shape[0]
    sub = pts[:: max(1, m // 500_000)]
    if sub.shape[0] >= 1000:
        k_lo = max(1, int(BOX_QUANTILE * sub.shape[0]))
        lo = torch.kthvalue(sub, k_lo, dim=0).values
        hi = torch.kthvalue(sub, sub.shape[0] - k_lo + 1, dim=0).values
    else:
        lo, hi = sub.amin(dim=0), sub.amax(dim=0)
    full = torch.stack([lo, hi, pts.amin(dim=0), pts.amax(dim=0)]).cpu()      # the read-back that sizes the grid
    lo, hi = full[0].tolist(), full[1].tolist()
    pad = [0.02 * max(h - l, 0.0) for l, h in zip(lo, hi)]
    lo = [max(l - p, fl) for l, p, fl in zip(lo, pad, full[2].tolist())]
    hi = [min(h + p, fh) for h, p, fh in zip(hi, pad, full[3].tolist())]
    max_abs = float(full.abs().max())
    return lo, hi, max_abs


def knn_mean_distance(points, k=NB_NEIGHBORS):
    """float64 [M]: mean distance of every point to its k nearest points of the cloud (itself included).

    Exact.  Level 0: a grid over the robust bounding box whose non-empty cells hold ~8 points answers every query whose
    k-th neighbour lies within SHELL_BUDGET shells of cells; the others (sparse regions, floaters) are asked again on
    grids with LEVEL_FACTOR x larger cells until the shells cover the whole grid."""
    L = nv.lib()
    pts = points.detach().to(torch.float32).contiguous()
    m, dev = pts.shape[0], pts.device
    if m == 0:
        return torch.zeros((0,), dtype=torch.float64, device=dev)
    lo_h, hi_h, max_abs = _robust_box(pts)
    extent = [max(h - l, 0.0) for l, h in zip(lo_h, hi_h)]
    longest = max(max(extent), 1e-30)
    max_abs = max(max_abs, longest)
    stream = nv.stream_handle(dev)

    ws_bytes = L.g2pc_outlier_grid_workspace(m)
    ws = nv.workspace(ws_bytes, dev)
    sorted_pos = torch.empty((m, 4), dtype=torch.float32, device=dev)
    occupied = torch.zeros((1,), dtype=torch.int32, device=dev)
    origin = (C.c_float * 3)(*lo_h)
    cell_start = None

    def build(cell):
        nonlocal cell_start
        dims_l = _dims(extent, cell)
        cells = dims_l[0] * dims_l[1] * dims_l[2]
        dims = (C.c_int32 * 3)(*dims_l)
        if cell_start is None or cell_start.numel() < cells + 2:
            cell_start = None
            cell_start = torch.empty((cells + 2,), dtype=torch.int32, device=dev)    # (the range kernel also leaves m at [cells + 1])
        nv.check(L.g2pc_outlier_grid_build(nv.ptr(pts), m, C.byref(origin), cell, C.byref(dims), nv.ptr(sorted_pos),
                                           nv.ptr(cell_start), nv.ptr(occupied), nv.ptr(ws), ws_bytes, stream),
                 "outlier_grid_build")
        return dims, dims_l

    # level 0: refine until the non-empty cells are sparsely populated (surfaces fill far fewer cells than volumes)
    res = max(1.0, round(m ** (1.0 / 3.0)))                              # cells along the longest edge
    for attempt in range(6):
        cell = longest / res
        while True:
            d = _dims(extent, cell)
            if d[0] * d[1] * d[2] <= MAX_CELLS:
                break
            res /= 1.26
            cell = longest / res
        dims, dims_l = build(cell)
        per_cell = m / max(int(occupied.item()), 1)
        finer = res * min(4.0, max(1.26, (per_cell / TARGET_POINTS_PER_CELL) ** 0.5))
        df = _dims(extent, longest / finer)
        if per_cell <= 2.0 * TARGET_POINTS_PER_CELL or df[0] * df[1] * df[2] > MAX_CELLS or attempt == 5:
            break
        res = finer

    avg = torch.empty((m,), dtype=torch.float64, device=dev)
    lists = [torch.empty((m,), dtype=torch.int32, device=dev), None]
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    queries, nq, level = None, m, 0
    LAST_STATS.clear()
    LAST_STATS.update(levels=[], points_per_cell=per_cell)
    while True:
        slack = 1e-5 * cell + 1e-6 * max_abs
        out_list = lists[level % 2]
        if out_list is None or out_list.numel() < nq:
            out_list = lists[level % 2] = torch.empty((nq,), dtype=torch.int32, device=dev)
        nv.check(L.g2pc_outlier_knn_mean_distance(nv.ptr(sorted_pos), nv.ptr(cell_start), m, C.byref(origin), cell,
                                                  C.byref(dims), int(k), slack, nv.ptr(pts), nv.ptr(queries), nq,
                                                  SHELL_BUDGET, nv.ptr(out_list), nv.ptr(count), nv.ptr(avg), stream),
                 "outlier_knn_mean_distance")
        left = int(count.item())
        LAST_STATS["levels"].append(dict(cell=cell, dims=list(dims_l), queries=nq, unresolved=left))
        if left == 0:
            break
        if max(dims_l) <= SHELL_BUDGET + 1:                                # the shells of every query covered the grid
            raise RuntimeError("outlier cascade did not terminate")
        queries, nq, level = out_list, left, level + 1
        cell *= LEVEL_FACTOR
        dims, dims_l = build(cell)
    return avg


def statistical_outlier_mask(avg, std_ratio):
    """Open3D RemoveStatisticalOutliers' selection on the per-point mean distances (float64, on the device)."""
    pos = avg > 0
    valid = avg.shape[0]
    if valid < 2 or not bool(pos.any()):
        return pos
    cloud_mean = avg[pos].sum() / valid
    std_dev = torch.sqrt(((avg[pos] - cloud_mean) ** 2).sum() / (valid - 1))
    return pos & (avg < cloud_mean + std_ratio * std_dev)


def clean_point_cloud(points, colours, normals, std_ratio=10, device="cuda:0"):
    """mesh_handler.py:89-94: -> (points float64, colours int32, normals float64) of the inliers, on the input's device
    (`device` is accepted for signature compatibility; the data never leaves the GPU it is on)."""
    avg = knn_mean_distance(points, NB_NEIGHBORS)
    keep = statistical_outlier_mask(avg, float(std_ratio))
    pts = points.detach().to(torch.float32).to(torch.float64)[keep]
    # mesh_handler.py:47,52,60: clamp -> int32 (truncation) -> /255 (float64) -> *255 -> int
    cols = torch.clamp(colours.detach(), min=0, max=255).to(torch.int32)
    cols = (cols.to(torch.float64) / 255 * 255).to(torch.int32)[keep]
    nrm = None if normals is None else normals.detach().to(torch.float64)[keep]
    return pts, cols, nrm


def open3d_available():
    try:
        import open3d  # noqa: F401
        return True
    except Exception:
        return False


def generate_mesh(points, colours, normals, output_path, depth=12, laplacian_iters=10, std_ratio=3):
    """mesh_handler.py:66-87 -- Open3D's Poisson reconstruction + Laplacian smoothing.  The meshing algorithm is Open3D's
    own (a third-party CPU library after the hot path): when Open3D is installed this delegates to it
    with the reference's parameters, otherwise it raises -- and `main()` refuses --generate_mesh up front."""
    if not open3d_available():
        raise NotImplementedError("--generate_mesh needs Open3D's Poisson surface reconstruction (not installed); "
                                  "convert_3dgs_to_pc still returns the surface point cloud it would consume")
    import numpy as np
    import open3d as o3d
    pcd = o3d.geometry.PointCloud()
    pcd.points = o3d.utility.Vector3dVector(points.detach().cpu().double().numpy())
    pcd.colors = o3d.utility.Vector3dVector(np.clip(colours.detach().cpu().double().numpy(), 0, 255) / 255.0)
    pcd.normals = o3d.utility.Vector3dVector(normals.detach().cpu().double().numpy())
    pcd, _ = pcd.remove_statistical_outlier(nb_neighbors=20, std_ratio=std_ratio)
    mesh, _ = o3d.geometry.TriangleMesh.create_from_point_cloud_poisson(pcd, depth=depth)
    mesh = mesh.filter_smooth_laplacian(number_of_iterations=laplacian_iters)
    mesh.compute_vertex_normals()
    o3d.io.write_triangle_mesh(output_path, mesh)
    return mesh
