"""Edge cases of the hot path through the emulator: degenerate sizes, nothing visible, everything culled, zero budgets,
huge quotas (wave-per-Gaussian sampling), masks that hide everything, NaN covariances."""
import numpy as np
import pytest
import torch

from emu_util import emu, emu_exp  # noqa: F401
import ref_gauss as RG
from np_philox import keyed_normals
from g2pc import ops
from g2pc.synth import make_scene, make_cameras


def test_single_gaussian_and_tiny_inputs(emu):
    sc = make_scene(1, 3)
    cov, cov6, nrm = ops.build_covariances(sc.scales, sc.rots, 1.0, want_cov6=True, want_normals=True)
    assert cov.shape == (1, 3, 3) and bool(ops.validate_covariances_(cov).all())
    mags = ops.gaussian_magnitudes(cov, sc.opacities)
    ppg64, ppg, stats = ops.distribute_points(mags, 37)
    assert ppg.tolist() == [37]
    out = ops.sample_pointcloud(sc.xyz, cov, sc.colours * 255, nrm, ppg, int(stats[3]), exact=True, std=2.0, attempts=100,
                                seed=1, want_index=True)
    assert out.points.shape[0] == 37 and torch.equal(out.points[0], sc.xyz[0])       # the mean comes first
    assert ops.exclusive_scan_u32(torch.zeros(0, dtype=torch.int32)).tolist() == [0]
    assert ops.compact_index(torch.zeros(5, dtype=torch.bool)).numel() == 0


def test_large_quota_uses_wave_sampling_and_matches_oracle(emu):
    """One dominant Gaussian gets hundreds of points (wave64 path), the rest get 0..few (lane path)."""
    sc = make_scene(40, 5)
    cov = RG.covariances(sc.scales, sc.rots)
    cov, keep = RG.validate_covariances(cov)
    w = sc.opacities.clone()
    w[7] = 400.0                                              # contributions as weights: Gaussian 7 dominates
    seed = 9
    ref = RG.generate_pointcloud(sc.xyz, cov, sc.colours * 255, RG.normals(sc.scales, sc.rots), w, 3000, std=2.0,
                                 exact=True, attempts=100,
                                 eps_fn=lambda g, a, k: keyed_normals(seed, g[:, None], a, np.arange(k)[None, :]))
    mags = ops.gaussian_magnitudes(cov, w)
    _, ppg, stats = ops.distribute_points(mags, 3000)
    assert int(ppg.max()) > 1000
    out = ops.sample_pointcloud(sc.xyz, cov, sc.colours * 255, RG.normals(sc.scales, sc.rots), ppg, int(stats[3]),
                                exact=True, std=2.0, attempts=100, seed=seed, want_index=True)
    assert np.array_equal(ppg.numpy(), ref["ppg"].numpy())
    assert out.points.shape == tuple(ref["points"].shape)
    assert float((out.points - ref["points"]).abs().max()) < 1e-5
    assert np.array_equal(out.gauss_index.numpy(), ref["gauss_index"].numpy())


def test_nothing_in_front_of_the_camera(emu):
    import gauss_render, camera_handler
    from gauss_handler import Gaussians
    sc = make_scene(300, 6)
    G = Gaussians(sc.xyz + torch.tensor([0.0, 0.0, 50.0]), sc.scales, sc.rots, sc.colours, sc.opacities)   # behind the rig
    tr, intr = make_cameras(1, width=96, height=64, focal=80.0)
    name = next(iter(tr))
    tr[name] = torch.eye(4).tolist()                       # camera at the origin looking down -z
    for kind in ("python", "cuda"):
        R = gauss_render.get_renderer(kind, G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05, calculate_surface_distance=(kind == "cuda"),
                                      surface_distance_std=2.0 if kind == "cuda" else None)
        img = R(camera_handler.get_camera(kind, torch.tensor(tr[name]), intr[name]))[0]
        assert float((img - 1.0).abs().max()) == 0.0                       # white background everywhere
        assert int(R.get_visible_gaussians().sum()) == 0
    # and the pipeline reports it the way the reference does
    import gauss_to_pc as g2p
    s = g2p.GaussPointCloudSettings("python", 1000, True, 2.0, 0, True, 0.0, None, None, True, 0.0, True, None, 3, False,
                                    0.05, None, False, True, "cpu")
    with pytest.raises(Exception, match="after culling is 0"):
        g2p.convert_gaussians_to_pc(G, tr, intr, None, s)


def test_mask_hiding_everything_and_partial_tiles(emu):
    import gauss_render, camera_handler
    from gauss_handler import Gaussians
    sc = make_scene(400, 7, scale_lo=0.02, scale_hi=0.08)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    tr, intr = make_cameras(1, width=50, height=37, focal=45.0)         # 50x37: partial 16x16 tiles on both edges
    name = next(iter(tr))
    R = gauss_render.get_renderer("cuda", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                  visible_gaussian_threshold=0.05)
    cam = camera_handler.get_camera("cuda", torch.tensor(tr[name]), intr[name], mask=torch.zeros((37, 50), dtype=torch.int32))
    colour, radii, invd, dep = R.forward(cam, return_per_camera=True)
    assert float(colour.abs().max()) == 0.0 and float(dep.abs().max()) == 0.0      # masked pixels are never written
    assert float(R.last["contributions"].max()) == 0.0 and int(R.get_visible_gaussians().sum()) == 0
    assert int((radii > 0).sum()) > 0                                              # preprocess still ran


def test_nan_covariance_rows_follow_the_reference(emu):
    cov = torch.eye(3).repeat(4, 1, 1) * 1e-4
    cov[2] = float("nan")
    ref_valid, ref_keep = RG.validate_covariances(cov)
    keep = ops.validate_covariances_(cov.clone())
    assert keep.tolist() == ref_keep.tolist()               # eigvals(NaN) <= eps is False -> kept, as in the reference


def test_wide_radix_digits_and_zero_budget(emu_exp):
    from g2pc import _native as nv
    rng = np.random.default_rng(4)
    keys = rng.integers(0, 2 ** 32, size=9000, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(9000, dtype=np.uint32)
    try:
        assert nv.experiments().g2pc_set_sort_tuning(11, 1 << 21) == 0
        ko, vo = ops.sort_pairs_u32(torch.from_numpy(keys.view(np.int32)), torch.from_numpy(vals.view(np.int32)), 0, 32)
    finally:
        nv.experiments().g2pc_set_sort_tuning(8, 1 << 21)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(vo.numpy().view(np.uint32), vals[order])
    # wide_digit_bits = 10: ONE pass for a 9- or 10-bit field (the tile sort of 257 - 1 024 leaves), passes of up to 8 otherwise
    for lo, hi in ((20, 29), (3, 13), (0, 32), (5, 12)):
        try:
            assert nv.experiments().g2pc_set_sort_tuning(10, 1 << 21) == 0
            ko, vo = ops.sort_pairs_u32(torch.from_numpy(keys.view(np.int32)), torch.from_numpy(vals.view(np.int32)), lo, hi)
        finally:
            nv.experiments().g2pc_set_sort_tuning(8, 1 << 21)
        field = (keys >> np.uint32(lo)) & np.uint32((1 << (hi - lo)) - 1 if hi - lo < 32 else 0xFFFFFFFF)
        order = np.argsort(field, kind="stable")
        assert np.array_equal(vo.numpy().view(np.uint32), vals[order]), (lo, hi)
        assert np.array_equal(ko.numpy().view(np.uint32), keys[order]), (lo, hi)
    sizes = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    p, _, st = ops.distribute_points(sizes, 1)             # budget smaller than the number of Gaussians
    assert p.tolist() == RG.distribute_points(sizes, 1).tolist()


def test_tile_overflow_is_split_and_reported(emu, monkeypatch):
    """More Gaussians in one leaf than max_gaussians_per_tile: the leaf is split as the reference's queue splits it
    (tests/test_emu_quadtree.py holds the parity checks); check_tile_load() tells the largest load met."""
    import gauss_render, camera_handler
    from gauss_handler import Gaussians
    sc = make_scene(600, 12, scale_lo=0.05, scale_hi=0.1)
    G = Gaussians(sc.xyz * 0.05, sc.scales, sc.rots, sc.colours, sc.opacities)          # everything in the centre
    tr, intr = make_cameras(1, width=96, height=64, focal=80.0)
    name = next(iter(tr))
    monkeypatch.setattr(gauss_render.GaussHipRenderer, "MAX_GAUSSIANS_PER_TILE", 100)
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances)
    R(camera_handler.get_camera("python", torch.tensor(tr[name]), intr[name]))
    R.get_gaussian_colours()
    assert R.split_leaves > 0 and R.check_tile_load() > 100


def test_cull_large_gaussians_drops_the_largest(emu):
    """gauss_handler.py:235-250 (intent; the reference's own line is a dtype error): the floor(n (1 - p)) smallest
    Gaussians by get_gaussian_magnitudes() survive, ties in index order."""
    import numpy as np
    import torch
    from gauss_handler import Gaussians
    from g2pc import ops
    from g2pc.synth import make_scene
    sc = make_scene(3000, 17)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    mags = G.get_gaussian_magnitudes().numpy()
    order = ops.argsort_f64_nonnegative(torch.from_numpy(mags)).numpy()
    assert np.array_equal(order, np.argsort(mags, kind="stable"))
    G.cull_large_gaussians(0.1)
    keep = G.filter_indices.numpy()
    assert int(keep.sum()) == 2700
    assert mags[keep].max() <= mags[~keep].min()
    kept = G.filter_gaussians()
    assert G.xyz.shape[0] == 2700 and torch.equal(G.select(torch.from_numpy(mags)), torch.from_numpy(mags[kept.numpy()]))
    # exact ties: the stable sort keeps index order
    t = torch.tensor([3.0, 1.0, 3.0, 0.0, 1.0, 3.0], dtype=torch.float64)
    assert ops.argsort_f64_nonnegative(t).tolist() == [3, 1, 4, 0, 2, 5]


def test_threshold_at_or_below_the_floor_takes_the_exact_blend(emu):
    """get_renderer's default visible_gaussian_threshold is 0.0 and the visibility test a strict `>`
    (/root/reference/gauss_render.py:249-252, :387, :467-468): Gaussians whose contributions all lie in (0, 2^-25) are visible
    there; the renderer drops its transmittance floor by itself whenever the threshold does not lie above it."""
    from render_checks import assert_hidden_behind_wall
    assert_hidden_behind_wall("cpu")


def test_deferred_validate_cull_equals_the_eager_one(emu, monkeypatch):
    """convert_gaussians_to_pc reads validate_covariances' culled-row count only after the sampling was queued
    (DEFER_VALIDATE_CULL): when rows ARE culled (ill-conditioned covariances) the cloud is sampled again from the filtered set
    and must equal the cloud of the eager order, row for row; and the magnitudes that reuse the validation's sqrt(area) must equal
    the ones from a fresh eigen-decomposition bit for bit."""
    import gauss_to_pc as g2p
    from gauss_handler import Gaussians
    sc = make_scene(600, 12, scale_lo=0.01, scale_hi=0.05)
    # rows the clamp rounds cannot repair (a dynamic range of the spectrum beyond fp32: gauss_handler.py:142-166 culls them)
    gen = torch.Generator().manual_seed(5)
    bad = {}
    for i in (5, 9, 17, 300, 599):
        q, _ = torch.linalg.qr(torch.randn((3, 3), generator=gen))
        m = q @ torch.diag(torch.tensor([1e5, 1e-3, -2e-3])) @ q.T
        bad[i] = (m + m.T) * 0.5
    s = g2p.GaussPointCloudSettings("python", 6000, True, 2.0, 0, False, 0.0, None, None, True, 0.0, True, None, 3, False,
                                    0.05, None, False, True, "cpu")
    clouds = []
    for defer in (False, True):
        monkeypatch.setattr(g2p, "DEFER_VALIDATE_CULL", defer)
        G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours.clone(), sc.opacities)
        cov = G.covariances.clone()
        for i, m in bad.items():
            cov[i] = m
        G.covariances = cov
        cloud, _ = g2p.convert_gaussians_to_pc(G, None, None, None, s, seed=7)
        assert G.last_validate_culled and G.xyz.shape[0] < 600           # rows really were culled
        clouds.append((cloud.points.clone(), cloud.colours.clone(), G.xyz.shape[0]))
        # the kept sqrt(area) against a fresh decomposition of the validated matrices
        fresh = ops.gaussian_magnitudes(G.covariances, G.opacities)
        assert torch.equal(G.get_gaussian_magnitudes(), fresh)
    assert clouds[0][2] == clouds[1][2]
    assert torch.equal(clouds[0][0], clouds[1][0]) and torch.equal(clouds[0][1], clouds[1][1])
