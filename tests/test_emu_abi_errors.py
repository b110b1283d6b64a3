"""Error behaviour of the C ABI (include/g2pc.h): negative G2PC_ERR_* codes, a thread-local message, nothing queued."""
import ctypes as C

import numpy as np
import torch

from emu_util import emu  # noqa: F401


def _p(t):
    return C.c_void_p(t.data_ptr())


def test_error_codes_and_messages(emu):
    from g2pc import _native as nv
    L = nv.lib()
    n = 1000
    keys = torch.randint(0, 1 << 20, (n,), dtype=torch.int32)
    vals = torch.arange(n, dtype=torch.int32)
    ko, vo, kt, vt = (torch.empty(n, dtype=torch.int32) for _ in range(4))
    ws = torch.empty(L.g2pc_sort_workspace(n), dtype=torch.uint8)
    # aliased ping-pong buffers (the bug class that once corrupted 1024-tile lists) -> G2PC_ERR_ARG
    rc = L.g2pc_sort_pairs_u32(_p(keys), _p(vals), _p(ko), _p(vo), _p(keys), _p(vt), n, 0, 20, _p(ws), ws.numel(), None)
    assert rc == -1 and b"distinct" in L.g2pc_last_error()
    # workspace too small -> G2PC_ERR_WORKSPACE, outputs untouched
    ko.fill_(-7)
    rc = L.g2pc_sort_pairs_u32(_p(keys), _p(vals), _p(ko), _p(vo), _p(kt), _p(vt), n, 0, 20, _p(ws), 16, None)
    assert rc == -2 and b"workspace" in L.g2pc_last_error() and int((ko != -7).sum()) == 0
    # a correct call still works afterwards and sorts stably
    rc = L.g2pc_sort_pairs_u32(_p(keys), _p(vals), _p(ko), _p(vo), _p(kt), _p(vt), n, 0, 20, _p(ws), ws.numel(), None)
    assert rc == 0
    order = np.argsort(keys.numpy(), kind="stable")
    assert np.array_equal(vo.numpy(), order)
    # NULL pointers / non-positive sizes
    cov = torch.zeros((4, 3, 3))
    assert L.g2pc_build_covariances(None, None, 1.0, 4, _p(cov), None, None, None, None) == -1
    assert L.g2pc_validate_covariances(_p(cov), 4, 1, 5e-7, 1e-7, 1e-8, 3, None, None) == -1     # no keep mask
    assert L.g2pc_validate_covariances(None, 0, 1, 5e-7, 1e-7, 1e-8, 3, None, None) == 0          # empty input is fine


def test_raster_argument_checks(emu):
    import gauss_render
    from g2pc import _native as nv
    L = nv.lib()
    lay_host = gauss_render.tiles.python_quadtree_layout(64, 48, 60, 2)
    lay = gauss_render._DeviceLayout(lay_host, torch.device("cpu"))
    cam = gauss_render._Camera()
    n = 16
    f = lambda *s: torch.zeros(s, dtype=torch.float32)
    i = lambda *s: torch.zeros(s, dtype=torch.int32)
    rec, rect, sidx, offs = f(n, 16), i(n), i(n), i(n + 1)
    key, cols, tilebuf = torch.zeros(n, dtype=torch.int64), f(n, 3), f(lay.total_pixels * 3)
    ws = torch.empty(1 << 20, dtype=torch.uint8)
    args = lambda slot: (C.byref(cam), C.byref(lay.c), n, 0, _p(rec), _p(rect), _p(sidx), _p(offs), None, None, slot, 0.0, _p(key),
                         _p(cols), _p(tilebuf), None, 7, 0, None, _p(ws), ws.numel(), None)
    assert L.g2pc_raster_back_py(*args(0)) == -1 and b"camera_slot" in L.g2pc_last_error()       # slots are 1..255
    assert L.g2pc_raster_back_py(*args(256)) == -1
    assert L.g2pc_raster_back_py(*args(1)) == 0                                                    # empty camera: fine
    # the product library blends layouts of 2 sub-blocks per wave only (ABI 6; 1 / 4 exist in experiments builds)
    lay.c.chunk_subblocks = 4
    assert L.g2pc_raster_back_py(*args(1)) == -4 and b"chunk_subblocks must be 2" in L.g2pc_last_error()
    lay.c.chunk_subblocks = 2
    # capture needs a real (non-default) stream
    assert L.g2pc_graph_capture_begin(None) == -1
    assert L.g2pc_graph_launch(None, None) == -1
    # k > 32 neighbours is outside the register-resident kNN
    import mesh_handler  # noqa: F401  (registers the prototypes)
    o3, d3 = (C.c_float * 3)(0, 0, 0), (C.c_int32 * 3)(1, 1, 1)
    sp, cs, avg, un, cnt = f(4, 4), i(2), torch.zeros(4, dtype=torch.float64), i(4), i(1)
    rc = L.g2pc_outlier_knn_mean_distance(_p(sp), _p(cs), 4, C.byref(o3), 1.0, C.byref(d3), 33, 0.0, None, None, 0, 3,
                                          _p(un), _p(cnt), _p(avg), None)
    assert rc == -4 and b"k must be" in L.g2pc_last_error()
