"""Capture-and-replay camera pipeline (g2pc_raster_camera_py inside g2pc_graph_capture_*) through the emulator:
the replayed graphs must leave exactly the state of the two-call, host-synchronised path."""
import numpy as np
import pytest
import torch

from emu_util import emu  # noqa: F401
from g2pc.synth import make_scene, make_cameras


def _render_all(pipelined, monkeypatch, headroom=None, ncam=6, subblocks=2, batch=4):
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    gauss_render.clear_context_pool()           # a pooled context would bring the capacity an earlier test learned
    monkeypatch.setattr(gauss_render, "CAMERA_BATCH", batch)
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", subblocks)
    monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", pipelined)
    monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 3)
    if headroom is not None:
        monkeypatch.setattr(gauss_render, "CAPACITY_HEADROOM", headroom)
        monkeypatch.setattr(gauss_render, "MIN_CAPACITY", 1)
    sc = make_scene(700, 77, scale_lo=0.01, scale_hi=0.07)
    transforms, intr = make_cameras(ncam, width=200, height=112, focal=170.0)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                  visible_gaussian_threshold=0.05)
    for name in transforms:
        cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name])
        R(cam, return_image=not pipelined)
    cols = R.get_gaussian_colours().numpy().copy()
    keys = R.best_key.numpy().copy()
    return keys, cols, list(R.last_stats), R


@pytest.mark.parametrize("batch,ncam", [(1, 6), (2, 8), (4, 6), (4, 11), (3, 9)])
def test_replayed_graphs_equal_two_call_path(emu, monkeypatch, batch, ncam):
    """Cameras replayed `batch` at a time (one launch sequence with grid.y = batch, short last batch included) leave the
    state of one camera at a time."""
    k0, c0, st0, _ = _render_all(False, monkeypatch, ncam=ncam)
    k1, c1, st1, R = _render_all(True, monkeypatch, ncam=ncam, batch=batch)
    assert np.array_equal(k0, k1)
    assert np.array_equal(c0, c1)
    assert sorted(st0) == sorted(st1)                      # same instance counts per camera
    assert R.slots and any(sl.graph for sl in R.slots)    # the graphs were really captured and replayed
    assert all(sl.batch == batch for sl in R.slots)


def test_cameras_over_capacity_are_rendered_again(emu, monkeypatch):
    """A graph whose buffers are too small skips the camera as a whole; the host notices and re-renders it."""
    k0, c0, st0, _ = _render_all(False, monkeypatch)
    import gauss_render
    k1, c1, st1, R = _render_all(True, monkeypatch, headroom=0.6)
    assert np.array_equal(k0, k1)
    assert np.array_equal(c0, c1)
    assert sorted(st0) == sorted(st1)
    assert R.rerendered >= 1                               # some camera really overflowed and was rendered again
    assert R.capacity > int(min(x[0] for x in st1) * 0.6)  # and the capacity grew


def test_packed_two_pixel_blend_in_graph(emu, monkeypatch):
    k0, c0, _, _ = _render_all(False, monkeypatch, subblocks=2, ncam=3)
    k1, c1, _, _ = _render_all(True, monkeypatch, subblocks=2, ncam=3)
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)


def test_replayed_graph_with_an_empty_camera(emu, monkeypatch):
    """A camera that sees nothing (instance count 0) in the middle of the replayed sequence: the capacity-sized launches
    must all fall through and leave the state of the other cameras untouched."""
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", 2)
    monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 2)
    sc = make_scene(500, 78, scale_lo=0.01, scale_hi=0.07)
    transforms, intr = make_cameras(4, width=160, height=90, focal=140.0)
    names = sorted(transforms)
    away = torch.tensor(transforms[names[1]]).clone()
    away[:3, 3] = away[:3, 3] * 10.0                      # ten times further out ...
    away[:3, :3] = away[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=away.dtype))   # ... and looking away
    cams = [torch.tensor(transforms[names[0]]), away, torch.tensor(transforms[names[2]]), torch.tensor(transforms[names[3]])]
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    res = []
    for pipelined in (False, True):
        monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", pipelined)
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        for c2w in cams:
            R(camera_handler.get_camera("python", c2w, intr[names[0]]), return_image=not pipelined)
        cols = R.get_gaussian_colours().numpy().copy()          # flushes the staged batch
        res.append((R.best_key.numpy().copy(), cols, [s[0] for s in R.last_stats]))
    assert 0 in res[0][2] and sorted(res[0][2]) == sorted(res[1][2])          # the empty camera really was empty
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_depth_pile_up_falls_back_to_the_radix_sort(emu, monkeypatch):
    """2 500 Gaussians on a sheet facing the camera: almost all depths fall into one of the depth bucket sort's 1 024
    range buckets (> 1 024 keys, its room at this size), the captured graph skips the camera and the host renders it again
    through the two-call path (radix depth sort).  Result = the synchronous path's, bit for bit."""
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", 2)
    monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 2)
    sc = make_scene(2500, 79, scale_lo=0.01, scale_hi=0.05)
    xyz = sc.xyz.clone()
    xyz[:, 2] = 0.0                                          # the sheet z = 0 ...
    xyz[:40, 2] = torch.linspace(-0.9, 0.9, 40)              # ... plus a few Gaussians that stretch the depth range
    transforms, intr = make_cameras(3, width=128, height=72, focal=112.0)
    eye = torch.eye(4)
    eye[2, 3] = 3.5                                          # looking down -z at the sheet, head on: one depth for all of it
    cams = [eye.clone(), eye.clone(), torch.tensor(transforms[sorted(transforms)[1]])]
    cams[1][0, 3] = 0.2
    G = Gaussians(xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    res = []
    for pipelined in (False, True):
        monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", pipelined)
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        for c2w in cams:
            R(camera_handler.get_camera("python", c2w, intr[sorted(intr)[0]]), return_image=not pipelined)
        cols = R.get_gaussian_colours().numpy().copy()         # flushes: pending re-renders happen here
        res.append((R.best_key.numpy().copy(), cols, R.rerendered))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[1][2] >= 1                                    # the pile-up really happened and was handled


def test_deferred_colour_buffers_are_bounded(emu, monkeypatch):
    """The per-camera colour buffers of the deferred resolve form a bounded ring: with room for 2 cameras the
    renderer resolves and recycles every 2 cameras and still leaves the state of the two-call path."""
    import gauss_render
    k0, c0, _, _ = _render_all(False, monkeypatch, ncam=7)
    monkeypatch.setattr(gauss_render, "DEFERRED_MIN", 1)
    monkeypatch.setattr(gauss_render, "DEFERRED_MAX", 2)
    k1, c1, _, R = _render_all(True, monkeypatch, ncam=7)
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)
    assert len(R.ctx.cam_tilebufs) <= 2 and not R.deferred


@pytest.mark.parametrize("variant", ["default", "unfused", "regs8"])
def test_crowded_depth_bucket_within_its_room(emu, monkeypatch, variant):
    """700 of 2 500 Gaussians on a sheet facing the camera: one depth bucket holds 513 .. 1 024 keys -- within its room, so the
    captured graph keeps the camera, but beyond what most buckets hold: the fused sort-and-emit kernel takes its largest
    register path (16 items per lane) or, in the build with half the registers (-DG2PC_BK_EMIT_RMAX=8), its LDS network.
    All builds (and the round-4 chain: scan + k_duplicate + k_resolve_count, -DG2PC_FUSED_EMIT=0) must leave the state of the
    two-call path bit for bit, without a re-render."""
    import gauss_render
    import camera_handler
    from emu_util import build_emu
    from gauss_handler import Gaussians
    from g2pc import _native as nv
    defs = {"default": ("", ""), "unfused": ("_unfused", "-DG2PC_FUSED_EMIT=0 -DG2PC_PREPROCESS_MULTI=0"),
            "regs8": ("_r8", "-DG2PC_BK_EMIT_RMAX=8")}[variant]
    saved = (nv._LIB, nv._EMULATED)
    if defs[0]:
        nv._inject_for_tests(build_emu(*defs))
    try:
        gauss_render.clear_context_pool()
        monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", 2)
        monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 2)
        sc = make_scene(2500, 81, scale_lo=0.01, scale_hi=0.05)
        xyz = sc.xyz.clone()
        xyz[:700, 2] = 0.0                                       # the sheet z = 0: one depth for 700 Gaussians seen head on
        transforms, intr = make_cameras(3, width=128, height=72, focal=112.0)
        eye = torch.eye(4)
        eye[2, 3] = 3.5
        cams = [eye.clone(), eye.clone(), torch.tensor(transforms[sorted(transforms)[1]]), eye.clone()]
        cams[1][0, 3] = 0.2
        cams[3][1, 3] = -0.1
        G = Gaussians(xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
        res = []
        for pipelined in (False, True):
            monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", pipelined)
            R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                          visible_gaussian_threshold=0.05)
            for c2w in cams:
                R(camera_handler.get_camera("python", c2w, intr[sorted(intr)[0]]), return_image=not pipelined)
            cols = R.get_gaussian_colours().numpy().copy()
            res.append((R.best_key.numpy().copy(), cols, R.rerendered))
            R.close()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[1][2] == 0                                    # no bucket beyond its room: every camera stayed in its graph
    finally:
        gauss_render.clear_context_pool()
        nv._LIB, nv._EMULATED = saved


@pytest.mark.parametrize("semantics", ["python"])      # (the native-semantics call at this size: tests/test_gpu_graph_pipeline.py, 1.5 M)
def test_more_than_a_million_gaussians_take_the_8192_bucket_instances(emu, monkeypatch, semantics):
    """1 048 576 < n <= 2 097 152 Gaussians select k_bk_hist_w<BK_MAX> (8 192 depth buckets: 64 KB + 16 B of LDS, a
    gfx950-only size), the 32-buckets-per-thread loop of k_bk_scan and the 8 192-bucket emission -- no fixture reaches them
    (configs[4] at 1.0 M takes <4096>).  1.1 M Gaussians, nine tenths of them behind the cameras so that the emulator only
    sorts and blends ~100 k: the fused camera call of both semantics must leave the state of the two-call (radix) path."""
    import gauss_render
    import camera_handler
    import gaussian_pointcloud_rasterization as gpr
    from gauss_handler import Gaussians
    gauss_render.clear_context_pool()
    monkeypatch.setattr(gauss_render, "CAMERA_BATCH", 2)
    monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 2)
    monkeypatch.setattr(gpr, "PIPELINE_STREAMS", 2)
    n = 1_100_000
    sc = make_scene(n, 83, scale_lo=0.001, scale_hi=0.004)
    xyz = sc.xyz.clone()
    far = torch.arange(n) % 10 != 0
    xyz[far] = xyz[far] * 0.5 + torch.tensor([0.0, 0.0, 40.0])            # behind every camera of the rig below (and out of range)
    transforms, intr = make_cameras(24, width=96, height=54, focal=82.0)
    # ONE camera on the +z side, looking down -z at the origin (the emulator walks 1.1 M fibres per head kernel: two cameras took
    # 130 s of the CPU suite; tests/test_gpu_graph_pipeline.py renders 1.5 M Gaussians from several cameras in both semantics)
    names = [k for k in sorted(transforms) if transforms[k][2][3] > 2.0][:1]
    assert len(names) == 1
    G = Gaussians(xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    res = []
    for pipelined in (False, True):
        monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", pipelined)
        monkeypatch.setattr(gpr, "PIPELINE_IN_EMULATOR", pipelined)
        R = gauss_render.get_renderer(semantics, G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        for nm in names:
            R(camera_handler.get_camera(semantics, torch.tensor(transforms[nm]), intr[nm]), return_image=not pipelined)
        cols = R.get_gaussian_colours().numpy().copy()
        contrib = R.get_total_gaussian_contributions().numpy().copy()
        res.append((cols, contrib, R.rerendered))
        if hasattr(R, "close"):
            R.close()
    assert int((res[0][1] > 0).sum()) > 10_000                           # the scene really was rendered
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[1][2] <= 1                                                # (the first pipelined camera may learn the capacity)
    gauss_render.clear_context_pool()
