"""
ctypes binding of libg2pc.so (the C ABI declared in include/g2pc.h).

The library is the product: it is built in-tree by ``__graft_entry__.build()`` /
``make -C 3dgs-to-pc_amd/g2pc/csrc`` with ``hipcc --offload-arch=gfx950`` and there is NO CPU or
PyTorch fallback -- ``lib()`` raises if the shared object is missing, and every wrapper refuses
host tensors.  (tests/ may inject the fiber-emulated build of the same sources through
``_inject_for_tests``; nothing in the package does.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libg2pc.so")
ABI_VERSION = 7

_LIB: Optional[C.CDLL] = None
_EMULATED = False

_vp, _i64, _i32, _f32, _u64, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_size_t

_PROTOS = {
    "g2pc_last_error": (C.c_char_p, []),
    "g2pc_abi_version": (C.c_int, []),
    "g2pc_selftest_wave_reduce": (C.c_int, [_vp, _vp, _i64, _vp]),
    "g2pc_scan_workspace": (_sz, [_i64]),
    "g2pc_scan_exclusive_u32": (C.c_int, [_vp, _vp, _i64, _vp, _sz, _vp]),
    "g2pc_sort_workspace": (_sz, [_i64]),
    "g2pc_bucket_sort_workspace": (_sz, [_i64]),
    "g2pc_bucket_sort_u32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "g2pc_sort_pairs_u32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _sz, _vp]),
    "g2pc_build_covariances": (C.c_int, [_vp, _vp, _f32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "g2pc_cull_mask": (C.c_int, [_vp, _vp, _i64, C.c_int, _f32, _vp, _vp, _vp, _vp]),
    "g2pc_compact_workspace": (_sz, [_i64]),
    "g2pc_compact_index": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "g2pc_gather_rows": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "g2pc_gather_rows_multi": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    "g2pc_pack_ply_vertices": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "g2pc_validate_covariances": (C.c_int, [_vp, _i64, C.c_int, _f32, _f32, _f32, C.c_int, _vp, _vp]),
    "g2pc_validate_covariances_counted": (C.c_int, [_vp, _i64, C.c_int, _f32, _f32, _f32, C.c_int, _vp, _vp, _vp]),
    "g2pc_gaussian_magnitudes": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "g2pc_distribute_points_workspace": (_sz, [_i64]),
    "g2pc_distribute_points": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "g2pc_bincount_i32": (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    "g2pc_sampler_plan_workspace": (_sz, [_i64]),
    "g2pc_sampler_plan": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "g2pc_sampler_count": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _u64, _u64, _vp,
                                     _vp, _vp, _vp]),
}
_PROTOS["g2pc_scatter_ones_u8"] = (C.c_int, [_vp, _i64, _vp, _i64, _vp])
_PROTOS["g2pc_sampler_partition"] = (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp])
_PROTOS["g2pc_sampler_bin_table_workspace"] = (_sz, [_i64])
_PROTOS["g2pc_sampler_bin_table"] = (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp])
_PROTOS["g2pc_sampler_scan_workspace"] = (_sz, [_i64, _i32])
_PROTOS["g2pc_sampler_scan_counts"] = (C.c_int, [_vp, _vp, _i64, _i32, _vp, _sz, _vp])
_PROTOS["g2pc_sampler_sections"] = (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i64, C.c_int, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_sampler_emit_rows"] = (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _u64, _u64, _vp, _vp,
                                               _i64, _vp, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_validate_covariances_area"] = (C.c_int, [_vp, _i64, C.c_int, _f32, _f32, _f32, C.c_int, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_gaussian_magnitudes_from_area"] = (C.c_int, [_vp, _vp, _i64, _vp, _vp])
_PROTOS["g2pc_sampler_stage_plan"] = (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp])
_PROTOS["g2pc_sampler_count_staged"] = (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _u64, _u64, _vp,
                                                  _vp, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_sampler_emit_rows_staged"] = (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _vp,
                                                      _i64, _vp, _vp, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_sampler_run_workspace"] = (_sz, [_i64, _i64, _i64, _i32, _i32, _i64, _i64])
_PROTOS["g2pc_sampler_run_sections_offset"] = (_sz, [_i64, _i64, _i64, _i32, _i32, _i64, _i64])
_PROTOS["g2pc_sampler_run"] = (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _i64, _i64, _i32, _i64, _i64,
                                         _f32, _i32, _u64, _u64, C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp])
_PROTOS["g2pc_eval_sh"] = (C.c_int, [_i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp])
_PROTOS["g2pc_build_covariance_2d"] = (C.c_int, [_vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _vp, _vp])
_PROTOS["g2pc_projection_ndc"] = (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp])
_PROTOS["g2pc_get_radius"] = (C.c_int, [_vp, _i64, _vp, _vp])
_PROTOS["g2pc_get_rect"] = (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp])
_PROTOS["g2pc_mahalanobis"] = (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp])
_PROTOS["g2pc_sample_mvn"] = (C.c_int, [_vp, _vp, _i64, _i32, _u64, _u64, _i32, _vp, _vp])
# rasteriser prototypes are appended by g2pc/_native_raster.py style additions below
_RASTER_PROTOS = {}
# Entry points of -DG2PC_EXPERIMENTS builds only (csrc/experiments/knobs.inl: tuning and diagnostic knobs of rounds 2-4).  The
# product library exports none of them; tools/ and bench.py's tuning flags reach them through experiments().
_EXPERIMENT_PROTOS = {
    "g2pc_debug_set_head_threads": (C.c_int, [C.c_int]),
    "g2pc_set_sort_tuning": (C.c_int, [C.c_int, _i64]),
    "g2pc_raster_debug_chunk_work": (C.c_int, [C.c_void_p]),
    "g2pc_debug_set_extra_launches": (C.c_int, [C.c_int]),
    "g2pc_set_depth_sort": (C.c_int, [C.c_int]),
    "g2pc_set_blend_variant": (C.c_int, [C.c_int]),
    "g2pc_debug_set_walk_cap": (C.c_int, [C.c_int]),
    "g2pc_debug_stream_create_cu_mask": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
}


class G2pcError(RuntimeError):
    pass


def _bind(lib: C.CDLL, optional=()):
    for name, (res, args) in {**_PROTOS, **_RASTER_PROTOS}.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in optional:
                continue
            raise G2pcError("libg2pc.so does not export %s (stale build? run __graft_entry__.build())" % name)
        fn.restype = res
        fn.argtypes = args
    return lib


def lib() -> C.CDLL:
    """The loaded HIP library.  Raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise G2pcError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        _LIB = _bind(C.CDLL(LIB_PATH))
        v = _LIB.g2pc_abi_version()
        if v != ABI_VERSION:
            raise G2pcError("libg2pc.so ABI %d != binding ABI %d" % (v, ABI_VERSION))
    return _LIB


def experiments() -> C.CDLL:
    """The loaded library IF it is an experiments build (libg2pc_exp.so through LIB_PATH, or the test suite's "_exp" emulator
    build): binds the knobs of csrc/experiments/knobs.inl.  Raises for the product library, which has none."""
    L = lib()
    for name, (res, args) in _EXPERIMENT_PROTOS.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise G2pcError("%s is not in this library: tuning / diagnostic knobs exist in -DG2PC_EXPERIMENTS builds only "
                            "(tools/experiments/build_variant.sh exp -DG2PC_EXPERIMENTS, then run through "
                            "tools/experiments/ab_lib.py 3dgs-to-pc_amd/g2pc/libg2pc_exp.so)" % name)
        fn.restype = res
        fn.argtypes = args
    return L


def has_experiments() -> bool:
    try:
        experiments()
        return True
    except G2pcError:
        return False


def _inject_for_tests(path: str):
    """tests/ only: route the wrappers to the fiber-emulated build of the same sources."""
    global _LIB, _EMULATED
    _LIB = _bind(C.CDLL(path))
    _EMULATED = True


def emulated() -> bool:
    return _EMULATED


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise G2pcError("g2pc kernels need contiguous tensors")
    if t.device.type != "cuda" and not _EMULATED:
        raise G2pcError("g2pc HIP kernels need tensors resident in HBM (got device %s); no CPU fallback" % t.device)
    return C.c_void_p(t.data_ptr())


def stream_handle(device) -> Optional[C.c_void_p]:
    if _EMULATED or torch.device(device).type != "cuda":
        return None
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().g2pc_last_error()
        raise G2pcError("%s failed (%d): %s" % (what or "g2pc call", rc, msg.decode() if msg else ""))


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)


# ---- optional per-region device timing (bench.py only): HIP events on the stream the kernels run on --------
import contextlib

PROFILE = None   # dict name -> list of (start_event, end_event) when enabled


@contextlib.contextmanager
def region(name: str, device=None, stream=None):
    """Brackets the kernels launched inside with HIP events on the stream they are launched on."""
    if PROFILE is None or _EMULATED:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    st = stream if stream is not None else torch.cuda.current_stream(device)
    a.record(st)
    try:
        yield
    finally:
        b.record(st)
        PROFILE.setdefault(name, []).append((a, b))


def profile_summary():
    """name -> (launch count, total ms); call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in (PROFILE or {}).items():
        # entries: (begin, end) torch events, or milliseconds already read from HIP events recorded inside a graph
        out[name] = (len(evs), sum(e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in evs))
    return out
