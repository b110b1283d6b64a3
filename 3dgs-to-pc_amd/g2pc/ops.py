"""
Host-side operators over the g2pc C ABI (include/g2pc.h): thin torch-tensor wrappers, one per entry
point, plus the stage orchestration of the point sampler.  PyTorch supplies device memory and the
current HIP stream only; every byte of arithmetic happens in libg2pc.so.
"""
from __future__ import annotations

from math import floor
from typing import List, NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import _native as nv


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------ primitives
def exclusive_scan_u32(values: torch.Tensor) -> torch.Tensor:
    """int32/uint32 [n] -> int32 [n+1] exclusive prefix sums (last entry = total)."""
    n = values.numel()
    out = torch.empty((n + 1,), dtype=torch.int32, device=values.device)
    ws_bytes = nv.lib().g2pc_scan_workspace(n)
    ws = nv.workspace(ws_bytes, values.device)
    nv.check(nv.lib().g2pc_scan_exclusive_u32(nv.ptr(values.contiguous()), nv.ptr(out), n, nv.ptr(ws), ws_bytes,
                                              nv.stream_handle(values.device)), "scan")
    return out


def sort_pairs_u32(keys: torch.Tensor, vals: torch.Tensor, bit_lo: int = 0, bit_hi: int = 32):
    """Stable LSD radix sort of int32-typed (bit pattern = u32) key/value pairs."""
    n = keys.numel()
    ko, vo, kt, vt = (torch.empty_like(keys) for _ in range(4))
    ws_bytes = nv.lib().g2pc_sort_workspace(n)
    ws = nv.workspace(ws_bytes, keys.device)
    nv.check(nv.lib().g2pc_sort_pairs_u32(nv.ptr(keys.contiguous()), nv.ptr(vals.contiguous()), nv.ptr(ko),
                                          nv.ptr(vo), nv.ptr(kt), nv.ptr(vt), n, bit_lo, bit_hi, nv.ptr(ws),
                                          ws_bytes, nv.stream_handle(keys.device)), "sort")
    return ko, vo


def argsort_f64_nonnegative(x: torch.Tensor) -> torch.Tensor:
    """Stable ascending argsort (int32) of non-negative float64 values: their IEEE bit patterns order like unsigned
    64-bit integers, sorted least-significant word first with two stable 32-bit radix sorts."""
    x = x.to(torch.float64).contiguous()
    n = x.shape[0]
    words = x.view(torch.int32).reshape(n, 2)                    # little endian: [:, 0] low word, [:, 1] high word
    idx = torch.arange(n, dtype=torch.int32, device=x.device)
    _, order_lo = sort_pairs_u32(words[:, 0].contiguous(), idx)
    hi_in_lo_order = gather_rows(words[:, 1].contiguous(), order_lo)
    _, order = sort_pairs_u32(hi_in_lo_order, order_lo)
    return order


def scatter_ones_u8(dst_u8: torch.Tensor, index_i32: torch.Tensor) -> torch.Tensor:
    """dst[index] = 1 (uint8 destination, int32 positions)."""
    m = index_i32.numel()
    if m:
        nv.check(nv.lib().g2pc_scatter_ones_u8(nv.ptr(index_i32.contiguous()), m, nv.ptr(dst_u8), dst_u8.numel(),
                                               nv.stream_handle(dst_u8.device)), "scatter_ones_u8")
    return dst_u8


# ------------------------------------------------------------------------------------------ geometry
def build_covariances(log_scales: torch.Tensor, rots: torch.Tensor, scaling_modifier: float = 1.0,
                      want_cov6: bool = False, want_normals: bool = False, want_rotmat: bool = False):
    """gauss_handler.py:26-63 (+ :89-106 normals, :12-24 strip_symmetric) in one pass."""
    s, q = _f32c(log_scales), _f32c(rots)
    n = s.shape[0]
    cov = torch.empty((n, 3, 3), dtype=torch.float32, device=s.device)
    cov6 = torch.empty((n, 6), dtype=torch.float32, device=s.device) if want_cov6 else None
    nrm = torch.empty((n, 3), dtype=torch.float32, device=s.device) if want_normals else None
    rm = torch.empty((n, 3, 3), dtype=torch.float32, device=s.device) if want_rotmat else None
    nv.check(nv.lib().g2pc_build_covariances(nv.ptr(s), nv.ptr(q), float(scaling_modifier), n, nv.ptr(cov),
                                             nv.ptr(cov6), nv.ptr(nrm), nv.ptr(rm), nv.stream_handle(s.device)),
             "build_covariances")
    if want_rotmat:
        return cov, cov6, nrm, rm
    return cov, cov6, nrm


def cull_mask_(mask_u8: torch.Tensor, xyz: Optional[torch.Tensor], opacities: Optional[torch.Tensor],
               min_opacity: Optional[float], bbox_min, bbox_max) -> torch.Tensor:
    """In-place gauss_handler.py:195-224 on a uint8 mask."""
    import ctypes as C
    n = mask_u8.numel()
    bmin = (C.c_float * 3)(*[float(v) for v in bbox_min]) if bbox_min is not None else None
    bmax = (C.c_float * 3)(*[float(v) for v in bbox_max]) if bbox_max is not None else None
    x = _f32c(xyz) if xyz is not None else None
    o = _f32c(opacities).reshape(-1) if opacities is not None else None
    nv.check(nv.lib().g2pc_cull_mask(nv.ptr(x), nv.ptr(o), n, 1 if min_opacity is not None else 0,
                                     float(min_opacity or 0.0), C.cast(bmin, C.c_void_p) if bmin else None,
                                     C.cast(bmax, C.c_void_p) if bmax else None, nv.ptr(mask_u8),
                                     nv.stream_handle(mask_u8.device)), "cull_mask")
    return mask_u8


def compact_index(mask: torch.Tensor) -> torch.Tensor:
    """Ascending indices (int32) of the set entries of a bool/uint8 mask (one host read-back for the count)."""
    m8 = mask.to(torch.uint8).contiguous()
    n = m8.numel()
    index = torch.empty((max(n, 1),), dtype=torch.int32, device=m8.device)
    count = torch.zeros((1,), dtype=torch.int32, device=m8.device)
    ws_bytes = nv.lib().g2pc_compact_workspace(n)
    ws = nv.workspace(ws_bytes, m8.device)
    nv.check(nv.lib().g2pc_compact_index(nv.ptr(m8), n, nv.ptr(index), nv.ptr(count), nv.ptr(ws), ws_bytes,
                                         nv.stream_handle(m8.device)), "compact_index")
    return index[:int(count.item())]


def gather_rows(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """src[index] for a contiguous tensor whose rows are a multiple of 4 bytes."""
    src = src.contiguous()
    m = index.numel()
    row_bytes = (src.numel() // max(src.shape[0], 1)) * src.element_size() if src.shape[0] else 0
    out = torch.empty((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if m == 0:
        return out
    if row_bytes % 4 != 0:            # bool / uint8 rows: widen, gather, narrow
        return gather_rows(src.to(torch.int32), index).to(src.dtype)
    nv.check(nv.lib().g2pc_gather_rows(nv.ptr(src), nv.ptr(index.contiguous()), m, row_bytes, nv.ptr(out),
                                       nv.stream_handle(src.device)), "gather_rows")
    return out


def gather_rows_multi(srcs, index: torch.Tensor):
    """[src[index] for src in srcs] (None entries pass through) in ONE launch per eight arrays; rows must be whole 4-byte
    words (others take gather_rows)."""
    import ctypes as C
    out = [None] * len(srcs)
    todo = []
    m = index.numel()
    for i, t in enumerate(srcs):
        if t is None:
            continue
        t = t.contiguous()
        rb = (t.numel() // max(t.shape[0], 1)) * t.element_size() if t.shape[0] else 0
        if m == 0 or rb == 0 or rb % 4 != 0:
            out[i] = gather_rows(t, index)
        else:
            out[i] = torch.empty((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            todo.append((t, out[i], rb))
    idx = index.contiguous()
    for k in range(0, len(todo), 8):
        part = todo[k:k + 8]
        n = len(part)
        S = (C.c_void_p * n)(*[p[0].data_ptr() for p in part])
        D = (C.c_void_p * n)(*[p[1].data_ptr() for p in part])
        R = (C.c_int32 * n)(*[p[2] for p in part])
        nv.ptr(part[0][0])                                      # (device check: raises for host tensors outside the emulator)
        nv.check(nv.lib().g2pc_gather_rows_multi(C.cast(S, C.c_void_p), C.cast(D, C.c_void_p), C.cast(R, C.c_void_p), n, nv.ptr(idx), m,
                                                 nv.stream_handle(idx.device)), "gather_rows_multi")
    return out


def validate_covariances_(cov: torch.Tensor, regularise: bool = True, reg_eps: float = 5e-7, eps: float = 1e-7,
                          min_eps: float = 1e-8, iters: int = 3, want_count: bool = False, want_area: bool = False,
                          defer_count: bool = False):
    """In-place gauss_handler.py:142-166; returns the keep mask (bool[n]); want_count=True also returns the number of
    culled rows as a python int (ONE 4-byte read-back -- the caller's `if anything was culled`), or with defer_count the
    device tensor holding it (no read-back).  want_area=True appends sqrt(ellipsoid area) f32[n] of the validated matrices
    (gaussian_magnitudes_from_area: the magnitudes without a second eigen-decomposition)."""
    assert cov.dtype == torch.float32 and cov.is_contiguous()
    n = cov.shape[0]
    keep = torch.empty((n,), dtype=torch.uint8, device=cov.device)
    count = torch.zeros((1,), dtype=torch.int32, device=cov.device) if want_count else None
    area = torch.empty((n,), dtype=torch.float32, device=cov.device) if want_area else None
    nv.check(nv.lib().g2pc_validate_covariances_area(nv.ptr(cov), n, int(regularise), reg_eps, eps, min_eps, iters,
                                                     nv.ptr(keep), nv.ptr(count), nv.ptr(area), nv.stream_handle(cov.device)),
             "validate_covariances")
    keep = keep.view(torch.bool)                       # 0 / 1 bytes: a reinterpretation, not a conversion kernel
    out = (keep,)
    if want_count:
        out += (count if defer_count else int(count.item()),)
    if want_area:
        out += (area,)
    return out if len(out) > 1 else keep


def gaussian_magnitudes_from_area(sqrt_area: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:278-279 on the sqrt(area) validate_covariances_ kept -> float64 [n]."""
    a, w = _f32c(sqrt_area), _f32c(weights).reshape(-1)
    n = a.shape[0]
    assert w.shape[0] == n
    out = torch.empty((n,), dtype=torch.float64, device=a.device)
    nv.check(nv.lib().g2pc_gaussian_magnitudes_from_area(nv.ptr(a), nv.ptr(w), n, nv.ptr(out), nv.stream_handle(a.device)),
             "gaussian_magnitudes_from_area")
    return out


def gaussian_magnitudes(cov: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """gauss_handler.py:252-279 -> float64 [n]."""
    c, w = _f32c(cov), _f32c(weights).reshape(-1)
    n = c.shape[0]
    out = torch.empty((n,), dtype=torch.float64, device=c.device)
    nv.check(nv.lib().g2pc_gaussian_magnitudes(nv.ptr(c), nv.ptr(w), n, nv.ptr(out), nv.stream_handle(c.device)),
             "gaussian_magnitudes")
    return out


# ------------------------------------------------------------------------------------------ allocation
def distribute_points(sizes: torch.Tensor, num_points: int):
    """gauss_to_pc.py:73-90.  Returns (ppg float64[n], ppg int32[n], stats int64[4] on device)."""
    s = sizes.to(torch.float64).contiguous()
    n = s.shape[0]
    ppg64 = torch.empty_like(s)
    ppg32 = torch.empty((n,), dtype=torch.int32, device=s.device)
    stats = torch.empty((4,), dtype=torch.int64, device=s.device)
    ws_bytes = nv.lib().g2pc_distribute_points_workspace(n)
    ws = nv.workspace(ws_bytes, s.device)
    nv.check(nv.lib().g2pc_distribute_points(nv.ptr(s), n, int(num_points), nv.ptr(ppg64), nv.ptr(ppg32),
                                             nv.ptr(stats), nv.ptr(ws), ws_bytes, nv.stream_handle(s.device)),
             "distribute_points")
    return ppg64, ppg32, stats


def bincount(values_i32: torch.Tensor, length: int) -> torch.Tensor:
    hist = torch.zeros((length,), dtype=torch.int32, device=values_i32.device)
    nv.check(nv.lib().g2pc_bincount_i32(nv.ptr(values_i32.contiguous()), values_i32.numel(), nv.ptr(hist), length,
                                        nv.stream_handle(values_i32.device)), "bincount")
    return hist


def calculate_bin_sizes_from_hist(hist: np.ndarray) -> Tuple[int, int]:
    """gauss_to_pc.py:105-138 on the host-side histogram of points-per-Gaussian."""
    dist = hist[hist != 0]
    g2 = np.absolute(np.gradient(np.gradient(dist)))
    bin_size = max(len(dist) // 100, 1)
    g2 = g2[:len(g2) - len(g2) % bin_size]
    sums = g2.reshape(-1, bin_size).sum(axis=1)
    cut = np.max(sums) // 50
    peak = int(np.argmax(sums))
    below = np.nonzero(sums[peak:] < cut)[0]
    start_bin = int(below[0]) if below.shape[0] != 0 else 1
    return start_bin, bin_size


def bin_table_from_hist(hist: np.ndarray, exact: bool) -> List[Tuple[float, float, int]]:
    """gauss_to_pc.py:308-337: (start, end, quota) per bin in loop order."""
    pd = np.nonzero(hist)[0].astype(np.float64)                      # torch.unique(points_per_gaussian)
    if not exact:
        start_bin, bin_size = calculate_bin_sizes_from_hist(hist)
        tail = np.unique(np.ceil(pd[start_bin:] / bin_size)) * bin_size
        pd = np.concatenate([pd[:start_bin], tail])
    out = []
    for i, s in enumerate(pd):
        e = pd[i + 1] if i != len(pd) - 1 else s + 1
        out.append((float(s), float(e), floor(s + (e - s) / 2)))
    return out


class SampledCloud(NamedTuple):
    points: torch.Tensor                 # f32 [m,3]
    colours: torch.Tensor                # f32 [m,3]
    normals: Optional[torch.Tensor]      # f32 [m,3]
    gauss_index: Optional[torch.Tensor]  # i32 [m] (index into the arrays handed in)
    bins: list
    emitted_per_attempt: list


WAVE_MODE_MIN_DRAWS = 32   # quota-1 at and above which one Gaussian is sampled by a whole wave64
HIST_GUESS = 8192          # histogram length used before max(points per Gaussian) is known on the host
ATTEMPT_CHUNK = 8
DRAW_ONCE = True           # the count pass keeps the points it may have to emit, the emission copies (G2pcSampleStage)
ONE_CALL_TAIL = True       # partition .. emission through g2pc_sampler_run (one library call, one workspace)


import ctypes as C  # noqa: E402


class _SampleStage(C.Structure):
    """G2pcSampleStage (include/g2pc.h)."""
    _fields_ = [("thread_rows", C.c_void_p), ("wave_rows", C.c_void_p), ("wave_row_start", C.c_void_p)]


def mahalanobis(means: torch.Tensor, samples: torch.Tensor, covs: torch.Tensor) -> torch.Tensor:
    m, s_, c = _f32c(means), _f32c(samples), _f32c(covs)
    out = torch.empty((m.shape[0],), dtype=torch.float32, device=m.device)
    nv.check(nv.lib().g2pc_mahalanobis(nv.ptr(m), nv.ptr(s_), nv.ptr(c), m.shape[0], nv.ptr(out),
                                       nv.stream_handle(m.device)), "mahalanobis")
    return out


def sample_mvn(means: torch.Tensor, covs: torch.Tensor, n: int, seed: int, gid_base: int = 0,
               attempt: int = 0) -> torch.Tensor:
    m, c = _f32c(means), _f32c(covs)
    out = torch.empty((n, m.shape[0], 3), dtype=torch.float32, device=m.device)
    nv.check(nv.lib().g2pc_sample_mvn(nv.ptr(m), nv.ptr(c), m.shape[0], int(n), int(seed), int(gid_base),
                                      int(attempt), nv.ptr(out), nv.stream_handle(m.device)), "sample_mvn")
    return out


_PINNED = {}


def _pinned_i64(device, n=2):
    """A small pinned host buffer per device that kernels write through its device mapping (read after one sync)."""
    key = (str(device), n)
    if key not in _PINNED:
        t = torch.zeros((n,), dtype=torch.int64)
        if torch.device(device).type == "cuda" and not nv.emulated():
            t = t.pin_memory()
        _PINNED[key] = t
    return _PINNED[key]


def sample_pointcloud(xyz: torch.Tensor, cov: torch.Tensor, colours: torch.Tensor,
                      normals: Optional[torch.Tensor], ppg_i32: torch.Tensor, max_ppg: Optional[int], *, exact: bool,
                      std: float, attempts: int, seed: int, gid_base: int = 0, want_index: bool = False,
                      bins: Optional[list] = None, emit_means: bool = True,
                      stats: Optional[torch.Tensor] = None) -> SampledCloud:
    """The bin loop of generate_pointcloud (gauss_to_pc.py:308-371) + create_new_gaussian_points
    (gauss_to_pc.py:157-275) for all bins at once, in the reference's output order.
    `bins` overrides the bin table ((start, end, quota) triples); emit_means=False drops the centre points
    (create_new_gaussian_points on its own).

    Host round trips: ONE to build the bin table from the points-per-Gaussian histogram (a few hundred numbers), and
    one at the very end for the number of points produced.  Everything between -- partition by bin, count pass, scans,
    section table, emission -- is queued without waiting (attempts <= ATTEMPT_CHUNK, i.e. the default binned mode);
    `exact_num_points` (100 attempts) reads the number of unfinished Gaussians back after every chunk of attempts."""
    L = nv.lib()
    dev = xyz.device
    st = nv.stream_handle(dev)
    xyz, cov, colours = _f32c(xyz), _f32c(cov), _f32c(colours)
    normals = _f32c(normals) if normals is not None else None
    G = xyz.shape[0]

    plan = None
    if bins is None and max_ppg is None and stats is not None:
        # The bin table is built ON THE DEVICE from the histogram (g2pc_sampler_bin_table: numpy's arithmetic value for
        # value); the host reads back ten numbers -- how many bins, how many Gaussians they hold, the bound of the output
        # rows -- through pinned memory after one stream synchronisation.  No histogram download, no numpy in the middle
        # of the job, no look-up-table upload.
        hist_dev = bincount(ppg_i32, HIST_GUESS)
        lut_d = torch.empty((HIST_GUESS,), dtype=torch.int32, device=dev)
        quota_d = torch.empty((HIST_GUESS,), dtype=torch.int32, device=dev)
        bin_start = torch.empty((HIST_GUESS + 2,), dtype=torch.int32, device=dev)
        bin_lo = torch.empty((HIST_GUESS,), dtype=torch.int32, device=dev)
        plan_host = _pinned_i64(dev, 12)
        tb = L.g2pc_sampler_bin_table_workspace(HIST_GUESS)
        tws = nv.workspace(tb, dev)
        stats64 = stats.to(torch.int64).contiguous()
        nv.check(L.g2pc_sampler_bin_table(nv.ptr(hist_dev), HIST_GUESS, nv.ptr(stats64), int(exact),
                                          1 if emit_means else 0, WAVE_MODE_MIN_DRAWS, nv.ptr(lut_d), nv.ptr(quota_d),
                                          nv.ptr(bin_start), nv.ptr(bin_lo), C_void(plan_host), nv.ptr(tws), tb, st),
                 "sampler_bin_table")
        if dev.type == "cuda" and not nv.emulated():
            torch.cuda.current_stream(dev).synchronize()                                   # round trip #1: ten numbers
        plan = [int(v) for v in plan_host.tolist()]
        if plan[6] == 2:
            raise ValueError("fewer than two distinct points-per-Gaussian values: the reference's calculate_bin_sizes "
                             "(np.gradient) cannot bin such a distribution")
        if plan[6] != 0:
            plan = None                                      # some Gaussian got >= HIST_GUESS points: the host path below
    if plan is not None:
        B, gv, p_wave, any_sampling, means_rows, rows_ub = plan[0], plan[1], plan[2], bool(plan[3]), plan[4], plan[5]
        lane_planes = plan[10]
        lut_len = HIST_GUESS
        bins = _LazyBins(bin_lo, quota_d, B)
    else:
        if max_ppg is None:
            # histogram with a generous fixed length, ONE read-back for it and distribute_points' stats (falls back to the
            # exact length if some Gaussian got more than HIST_GUESS points)
            hist_dev = bincount(ppg_i32, HIST_GUESS)
            both = torch.cat([stats.to(torch.int64), hist_dev.to(torch.int64)]).cpu().numpy()            # round trip #1
            max_ppg = int(both[3])
            hist = both[4:4 + max_ppg + 1] if max_ppg < HIST_GUESS else None
        else:
            hist = None
        if hist is None:
            hist = bincount(ppg_i32, int(max_ppg) + 1).cpu().numpy().astype(np.int64)      # round trip #1
        if bins is None:
            bins = bin_table_from_hist(hist, exact)
        B = len(bins)
        lut = np.full((int(max_ppg) + 1,), -1, dtype=np.int32)
        quota = np.zeros((max(B, 1),), dtype=np.int32)
        members = np.zeros((max(B, 1),), dtype=np.int64)
        for b, (s, e, n) in enumerate(bins):
            quota[b] = n
            lo, hi = int(np.ceil(s)), int(np.ceil(e))
            hi = min(hi, int(max_ppg) + 1)
            if n > 0 and hi > lo:
                lut[lo:hi] = b
                members[b] = hist[lo:hi].sum()
        # bin sizes are known on the host from the histogram: bin_start is uploaded with the other small tables (one copy)
        bs_host = np.zeros((B + 2,), dtype=np.int64)
        bs_host[1:B + 1] = np.cumsum(members[:B])
        bs_host[B + 1] = bs_host[B]
        gv = int(bs_host[B])
        packed = torch.from_numpy(np.concatenate([lut, quota, bs_host.astype(np.int32)])).to(dev)
        lut_d, quota_d = packed[:lut.shape[0]], packed[lut.shape[0]:lut.shape[0] + quota.shape[0]]
        bin_start = packed[lut.shape[0] + quota.shape[0]:]
        lut_len = lut.shape[0]
        wave_bins = [b for b in range(B) if quota[b] - 1 >= WAVE_MODE_MIN_DRAWS and members[b] > 0]
        p_wave = int(bs_host[wave_bins[0]]) if wave_bins else gv
        any_sampling = bool(np.any((quota[:B] > 1) & (members[:B] > 0))) if B else False
        means_rows = int(sum(members[b] for b in range(B) if quota[b] > 0)) if emit_means else 0
        rows_ub = means_rows + int(sum(members[b] * max(int(quota[b]) - 1, 0) for b in range(B)))
        lane_planes = max([int(quota[b]) - 1 for b in range(B) if members[b] > 0 and 0 < quota[b] - 1 < WAVE_MODE_MIN_DRAWS] or [0])

    def outputs(rows):
        pts = torch.empty((rows, 3), dtype=torch.float32, device=dev)
        cols = torch.empty((rows, 3), dtype=torch.float32, device=dev)
        nrm = torch.empty((rows, 3), dtype=torch.float32, device=dev) if normals is not None else None
        gidx = torch.empty((rows,), dtype=torch.int32, device=dev) if want_index else None
        return pts, cols, nrm, gidx

    if ONE_CALL_TAIL and DRAW_ONCE and attempts <= ATTEMPT_CHUNK and G > 0:
        # the whole tail -- partition, staged count, scans, section table, emission -- as ONE library call over ONE workspace
        # (g2pc_sampler_run): the same launches without a dozen interpreter-level calls and allocations between them
        A = attempts if (any_sampling and gv > 0) else 0
        wave_rows = max(rows_ub - means_rows, 0) if p_wave < gv else 0
        sizes = (G, gv, p_wave, B, A, max(int(lane_planes), 0), wave_rows)
        wb = L.g2pc_sampler_run_workspace(*sizes)
        ws = nv.workspace(wb, dev)
        info = _pinned_i64(dev)
        pts, cols, nrm, gidx = outputs(rows_ub)
        with nv.region("sampler_run", dev):
            nv.check(L.g2pc_sampler_run(nv.ptr(xyz), nv.ptr(cov), nv.ptr(colours), nv.ptr(normals), nv.ptr(ppg_i32), G,
                                        nv.ptr(lut_d), lut_len, nv.ptr(quota_d), nv.ptr(bin_start), B, gv, p_wave,
                                        WAVE_MODE_MIN_DRAWS, sizes[5], wave_rows, float(std), A, int(seed), int(gid_base),
                                        1 if emit_means else 0, rows_ub, nv.ptr(pts), nv.ptr(cols), nv.ptr(nrm), nv.ptr(gidx),
                                        C_void(info), nv.ptr(ws), wb, st), "sampler_run")
        if dev.type == "cuda" and not nv.emulated():
            torch.cuda.current_stream(dev).synchronize()             # round trip #2: how many points came out
        M = int(info[0]) if B > 0 else 0
        assert 0 <= M <= rows_ub, (M, rows_ub)
        so = L.g2pc_sampler_run_sections_offset(*sizes)
        sec_base = ws[so:so + (max(B, 1) * (1 + A) + 1) * 8].view(torch.int64)
        return SampledCloud(pts[:M], cols[:M], nrm[:M] if nrm is not None else None, gidx[:M] if gidx is not None else None,
                            bins, _LazyPerAttempt(sec_base, B, A))

    perm = torch.empty((G,), dtype=torch.int32, device=dev)
    pbin = torch.empty((G,), dtype=torch.int32, device=dev)
    ws_bytes = L.g2pc_sampler_plan_workspace(G)
    ws = nv.workspace(ws_bytes, dev)
    nv.check(L.g2pc_sampler_partition(nv.ptr(ppg_i32), G, nv.ptr(lut_d), lut_len, B, nv.ptr(perm), nv.ptr(pbin),
                                      nv.ptr(ws), ws_bytes, st), "sampler_partition")
    added = torch.zeros((max(gv, 1) + 1,), dtype=torch.int32, device=dev)        # [gv] = unfinished Gaussians ("remaining")
    remaining = added[max(gv, 1):]
    info = _pinned_i64(dev)

    def sync():
        if dev.type == "cuda" and not nv.emulated():
            torch.cuda.current_stream(dev).synchronize()

    # ---- count pass: d[attempt][position]; exact mode loops over chunks of attempts until nothing is left unfinished ----
    # DRAW_ONCE: the count pass keeps every point it may have to emit (G2pcSampleStage, include/g2pc.h) and the emission copies
    # -- each keyed draw (Philox + Box-Muller + Cholesky, ~250 instructions) is evaluated once instead of twice
    stage = None
    if DRAW_ONCE and any_sampling and gv > 0:
        planes = max(int(lane_planes), 0)                         # the largest quota - 1 of a lane-mode bin (< WAVE_MODE_MIN_DRAWS)
        wave_rows = max(rows_ub - means_rows, 0) if p_wave < gv else 0
        stage = _SampleStage()
        stage.keep = (torch.empty((planes * max(p_wave, 0) * 3 + 4,), dtype=torch.float32, device=dev),
                      torch.empty((wave_rows * 3 + 4,), dtype=torch.float32, device=dev),
                      torch.empty((B + 1,), dtype=torch.int64, device=dev))
        stage.thread_rows, stage.wave_rows, stage.wave_row_start = (C_void(t) for t in stage.keep)
        if p_wave < gv:
            nv.check(L.g2pc_sampler_stage_plan(nv.ptr(bin_start), nv.ptr(quota_d), B, WAVE_MODE_MIN_DRAWS,
                                               nv.ptr(stage.keep[2]), st), "sampler_stage_plan")
    counts, befores = [], []            # dcount / have_before chunks, each [na, gv]
    a0 = 0
    while any_sampling and a0 < attempts:
        na = min(ATTEMPT_CHUNK, attempts - a0)
        dcount = torch.empty((na, gv), dtype=torch.int32, device=dev)
        if a0 > 0:
            remaining.zero_()
        with nv.region("sampler_count", dev):
            if stage is not None:
                hb = torch.empty((na, gv), dtype=torch.int32, device=dev)
                befores.append(hb)
                nv.check(L.g2pc_sampler_count_staged(nv.ptr(xyz), nv.ptr(cov), nv.ptr(perm), nv.ptr(pbin), nv.ptr(quota_d),
                                                     nv.ptr(bin_start), gv, p_wave, float(std), a0, na, int(seed), int(gid_base),
                                                     nv.ptr(added), nv.ptr(dcount), nv.ptr(hb), nv.ptr(remaining),
                                                     C.byref(stage), st), "sampler_count_staged")
            else:
                nv.check(L.g2pc_sampler_count(nv.ptr(xyz), nv.ptr(cov), nv.ptr(perm), nv.ptr(pbin), nv.ptr(quota_d), gv,
                                              p_wave, float(std), a0, na, int(seed), int(gid_base), nv.ptr(added),
                                              nv.ptr(dcount), nv.ptr(remaining), st), "sampler_count")
        counts.append(dcount)
        a0 += na
        if a0 < attempts:                                        # only exact mode gets here: is anybody still short?
            if int(remaining.cpu()[0]) == 0:
                break
    A = a0 if counts else 0
    dcount = counts[0] if len(counts) == 1 else (torch.cat(counts, 0) if counts else None)
    have_before = (befores[0] if len(befores) == 1 else torch.cat(befores, 0)) if befores else None
    dscan = torch.empty((A, gv + 1), dtype=torch.int32, device=dev) if A else None
    if A:
        sb = L.g2pc_sampler_scan_workspace(gv, A)
        sws = nv.workspace(sb, dev)
        nv.check(L.g2pc_sampler_scan_counts(nv.ptr(dcount), nv.ptr(dscan), gv, A, nv.ptr(sws), sb, st), "sampler_scan_counts")
    sec_base = torch.empty((max(B, 1) * (1 + A) + 1,), dtype=torch.int64, device=dev)
    nv.check(L.g2pc_sampler_sections(nv.ptr(bin_start), nv.ptr(quota_d), B, A, nv.ptr(dscan), gv, 1 if emit_means else 0,
                                     nv.ptr(sec_base), C_void(info), nv.ptr(remaining), st), "sampler_sections")
    pts, cols, nrm, gidx = outputs(rows_ub)
    if rows_ub > 0 and gv > 0 and B > 0:
        with nv.region("sampler_emit", dev):
            if stage is not None and A:
                nv.check(L.g2pc_sampler_emit_rows_staged(nv.ptr(xyz), nv.ptr(cov), nv.ptr(colours), nv.ptr(normals), nv.ptr(perm),
                                                         nv.ptr(bin_start), nv.ptr(quota_d), B, A, gv, p_wave, nv.ptr(dscan),
                                                         nv.ptr(have_before), nv.ptr(sec_base), rows_ub, C.byref(stage),
                                                         nv.ptr(pts), nv.ptr(cols), nv.ptr(nrm), nv.ptr(gidx), st),
                         "sampler_emit_rows_staged")
            else:
                nv.check(L.g2pc_sampler_emit_rows(nv.ptr(xyz), nv.ptr(cov), nv.ptr(colours), nv.ptr(normals), nv.ptr(perm),
                                                  nv.ptr(bin_start), B, 0, A, gv, int(seed), int(gid_base), nv.ptr(dscan),
                                                  nv.ptr(sec_base), rows_ub, nv.ptr(pts), nv.ptr(cols), nv.ptr(nrm),
                                                  nv.ptr(gidx), st), "sampler_emit_rows")
    sync()                                                        # round trip #2: how many points came out
    M = int(info[0]) if B > 0 else 0
    assert 0 <= M <= rows_ub, (M, rows_ub)
    pts, cols = pts[:M], cols[:M]
    nrm = nrm[:M] if nrm is not None else None
    gidx = gidx[:M] if gidx is not None else None
    return SampledCloud(pts, cols, nrm, gidx, bins, _LazyPerAttempt(sec_base, B, A))


def C_void(t: torch.Tensor):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


class _LazyBins:
    """The bin table ((start, end, quota) triples in loop order, gauss_to_pc.py:308-337) of a device-built plan: read from
    the device on first use (diagnostics and tests; the pipeline never looks at it)."""

    def __init__(self, bin_lo, quota, B):
        self._lo, self._q, self._B, self._v = bin_lo, quota, B, None

    def _get(self):
        if self._v is None:
            lo = self._lo[:self._B].cpu().numpy().astype(np.float64)
            q = self._q[:self._B].cpu().numpy()
            self._v = [(float(lo[i]), float(lo[i + 1] if i != self._B - 1 else lo[i] + 1), int(q[i])) for i in range(self._B)]
        return self._v

    def __getitem__(self, i):
        return self._get()[i]

    def __len__(self):
        return self._B

    def __iter__(self):
        return iter(self._get())

    def __eq__(self, other):
        return list(self._get()) == list(other)


class _LazyPerAttempt:
    """emitted rows per attempt (diagnostics): read from the device section table on first use."""

    def __init__(self, sec_base, B, A):
        self._t, self._B, self._A, self._v = sec_base, B, A, None

    def _get(self):
        if self._v is None:
            sb = self._t.cpu().numpy()
            size = (sb[1:] - sb[:-1]).reshape(max(self._B, 1), 1 + self._A) if self._B else np.zeros((1, 1 + self._A), np.int64)
            per = [int(size[:, 1 + a].sum()) for a in range(self._A)]
            while per and per[-1] == 0:          # trailing attempts that emitted nothing (the reference's loop would have stopped)
                per.pop()
            self._v = per
        return self._v

    def __getitem__(self, i):
        return self._get()[i]

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())
