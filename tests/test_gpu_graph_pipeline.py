"""Capture-and-replay camera pipeline on the MI355X: hipGraph replays on 4 streams against the two-call path, capacity
overflow and recapture, more cameras than the 8-bit order field, resolution changes mid-stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n=60_000, seed=5):
    from g2pc.synth import make_scene
    from gauss_handler import Gaussians
    sc = make_scene(n, seed, device="cuda:0", scale_lo=0.004, scale_hi=0.03)
    return Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)


def _render(G, cams, pipelined, headroom=None, min_capacity=None):
    import gauss_render
    import camera_handler
    old = (gauss_render.CAPACITY_HEADROOM, gauss_render.MIN_CAPACITY)
    if headroom is not None:
        gauss_render.CAPACITY_HEADROOM, gauss_render.MIN_CAPACITY = headroom, min_capacity
    try:
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        for tr, intr, res in cams:
            cam = camera_handler.get_camera("python", torch.tensor(tr), intr, colour_resolution=res)
            R(cam, return_image=not pipelined)
        cols = R.get_gaussian_colours().cpu().numpy()
        keys = R.best_key.cpu().numpy()
        return keys, cols, R
    finally:
        gauss_render.CAPACITY_HEADROOM, gauss_render.MIN_CAPACITY = old


def _cams(k, res=None, width=640, height=360):
    from g2pc.synth import make_cameras
    tr, intr = make_cameras(k, width=width, height=height, focal=550.0)
    return [(tr[name], intr[name], res) for name in sorted(tr)]


def test_graph_replays_equal_two_call_path():
    G = _scene()
    cams = _cams(12)
    k0, c0, _ = _render(G, cams, False)
    k1, c1, R = _render(G, cams, True)
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)
    assert R.slots and all(bool(sl.graph) for sl in R.slots) and R.rerendered == 0


def test_overflowing_cameras_are_rendered_again_and_capacity_grows():
    G = _scene()
    cams = _cams(10)
    k0, c0, _ = _render(G, cams, False)
    k1, c1, R = _render(G, cams, True, headroom=0.7, min_capacity=1)
    assert R.rerendered >= 1
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)


def test_resolution_change_and_more_cameras_than_order_slots():
    G = _scene(20_000, 9)
    cams = _cams(130, width=320, height=180) + _cams(130, res=256, width=320, height=180)    # 260 cameras, two layouts
    k0, c0, _ = _render(G, cams, False)
    k1, c1, R = _render(G, cams, True)
    # the order field wraps at 255 cameras (keys are rebased): contributions and winners' colours must still agree
    assert np.array_equal(k0 >> 32, k1 >> 32) and np.array_equal(c0, c1)


def test_wide_tile_field_and_its_shorter_camera_epoch():
    """16 384 quad-tree leaves (512 x 288 at max_tile_size 4 -- what 7680 x 4320 is at the default 60): the keys' tile field
    widens to 14 bits, which leaves 63 cameras per key epoch -- 70 cameras wrap it (keys rebased).  Replayed graphs against
    the two-call path."""
    import gauss_render
    old = gauss_render.GaussHipRenderer.MAX_TILE_SIZE
    gauss_render.GaussHipRenderer.MAX_TILE_SIZE = 4
    try:
        G = _scene(20_000, 9)
        cams = _cams(70, width=512, height=288)
        k0, c0, R0 = _render(G, cams, False)
        k1, c1, R = _render(G, cams, True)
    finally:
        gauss_render.GaussHipRenderer.MAX_TILE_SIZE = old
        gauss_render.clear_context_pool()
    assert R.seq_bits == 14 and R.camera_epoch == 63 and R0.seq_bits == 14
    assert np.array_equal(k0 >> 32, k1 >> 32) and np.array_equal(c0, c1)


@pytest.mark.parametrize("semantics", ["python", "cuda"])
def test_one_and_a_half_million_gaussians_take_the_8192_bucket_instances(semantics):
    """1 048 576 < n <= 2 097 152 Gaussians select k_bk_hist_w<BK_MAX> (8 192 depth buckets, 64 KB + 16 B of LDS: a
    gfx950 size), the 32-buckets-per-thread loop of k_bk_scan and the 8 192-bucket emission; the 1.0 M fixtures take <4096>.
    The fused camera call of both semantics at 1.5 M Gaussians against the two-call (radix) path: the same state bit for bit."""
    import gauss_render
    import camera_handler
    from g2pc.synth import make_scene, make_cameras
    from gauss_handler import Gaussians
    gauss_render.clear_context_pool()
    sc = make_scene(1_500_000, 31, device="cuda:0")
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    tr, intr = make_cameras(6, width=640, height=360, focal=550.0)
    res = []
    for pipelined in (False, True):
        R = gauss_render.get_renderer(semantics, G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        for nm in sorted(tr):
            R(camera_handler.get_camera(semantics, torch.tensor(tr[nm]), intr[nm]), return_image=not pipelined)
        res.append((R.get_gaussian_colours().cpu().numpy(), R.get_total_gaussian_contributions().cpu().numpy(), R.rerendered))
        if hasattr(R, "close"):
            R.close()
        del R
    gauss_render.clear_context_pool()
    assert int((res[0][1] > 0).sum()) > 100_000
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])
    assert res[1][2] <= 1
