"""Statistical outlier removal (--clean_pointcloud): grid kNN kernels through the emulator against the
CPU restatement of Open3D's algorithm (oracle/ref_clean.py, cKDTree)."""
import numpy as np
import pytest
import torch

from emu_util import emu  # noqa: F401


def _cloud(m, seed, outliers=6):
    rng = np.random.default_rng(seed)
    # a bumpy sheet (what a sampled surface looks like) + a blob + a few far-away floaters
    u = rng.uniform(-1, 1, size=(m, 2))
    sheet = np.stack([u[:, 0], u[:, 1], 0.2 * np.sin(3 * u[:, 0]) * np.cos(2 * u[:, 1]) + 0.01 * rng.normal(size=m)], 1)
    blob = rng.normal(size=(m // 4, 3)) * 0.05 + np.array([0.3, -0.2, 0.6])
    far = rng.uniform(-1, 1, size=(outliers, 3)) * np.array([6.0, 5.0, 7.0]) + np.array([0, 0, 9.0])
    pts = np.concatenate([sheet, blob, far]).astype(np.float32)
    pts[7] = pts[3]                                              # an exact duplicate pair
    return pts[rng.permutation(pts.shape[0])]


def test_knn_mean_distance_matches_kdtree(emu):
    import mesh_handler
    import ref_clean
    pts = _cloud(2400, 5)
    avg = mesh_handler.knn_mean_distance(torch.from_numpy(pts)).numpy()
    ref = ref_clean.knn_mean_distance(pts)
    assert avg.dtype == np.float64
    np.testing.assert_allclose(avg, ref, rtol=1e-14, atol=0)
    assert float((avg == ref).mean()) > 0.99                     # bit-identical but for the order of equal distances


@pytest.mark.parametrize("m,k", [(5, 20), (1, 20), (40, 32), (300, 7)])
def test_small_clouds_and_other_k(emu, m, k):
    import mesh_handler
    import ref_clean
    rng = np.random.default_rng(m)
    pts = rng.normal(size=(m, 3)).astype(np.float32)
    avg = mesh_handler.knn_mean_distance(torch.from_numpy(pts), k).numpy()
    np.testing.assert_allclose(avg, ref_clean.knn_mean_distance(pts, k), rtol=1e-14, atol=1e-300)


def test_clean_point_cloud_matches_restatement(emu):
    import mesh_handler
    import ref_clean
    pts = _cloud(2000, 9, outliers=5)
    rng = np.random.default_rng(1)
    cols = rng.uniform(-20, 280, size=(pts.shape[0], 3))         # out-of-range and fractional values: clamp + truncation
    cols[:300] = rng.integers(0, 256, size=(300, 3))
    nrm = rng.normal(size=(pts.shape[0], 3))
    p, c, n = mesh_handler.clean_point_cloud(torch.from_numpy(pts), torch.from_numpy(cols), torch.from_numpy(nrm), std_ratio=3)
    rp, rc, rn, keep, _ = ref_clean.clean_point_cloud(pts, cols, nrm, std_ratio=3)
    assert 0 < (~keep).sum() < 40                                 # the floaters (and little else) are gone
    assert p.dtype == torch.float64 and c.dtype == torch.int32 and n.dtype == torch.float64
    assert np.array_equal(p.numpy(), rp) and np.array_equal(c.numpy(), rc) and np.array_equal(n.numpy(), rn)
    assert np.array_equal(ref_clean.colour_round_trip(np.arange(256.0)), np.arange(256))   # /255*255 is exact on 0..255


def test_cli_clean_pointcloud(emu, tmp_path):
    import json
    import gauss_to_pc as g2p
    import gauss_dataloader as gd
    from g2pc.synth import make_scene, make_cameras
    from test_emu_io_cli import _write_3dgs_ply
    sc = make_scene(700, 8, scale_lo=0.01, scale_hi=0.05)
    _write_3dgs_ply(tmp_path / "scene.ply", sc)
    tr, intr = make_cameras(2, width=160, height=90, focal=140.0)
    frames = [{"file_path": "%s.png" % k, "transform_matrix": tr[k]} for k in tr]
    (tmp_path / "transforms.json").write_text(json.dumps({"w": 160, "h": 90, "fl_x": 140.0, "frames": frames}))
    args = ["--input_path", str(tmp_path / "scene.ply"), "--transform_path", str(tmp_path / "transforms.json"),
            "--renderer_type", "python", "--num_points", "6000", "--colour_quality", "original", "--quiet"]
    g2p.main(args + ["--output_path", str(tmp_path / "raw.ply")])
    g2p.main(args + ["--output_path", str(tmp_path / "clean.ply"), "--clean_pointcloud"])
    raw, clean = gd.read_ply_vertices(str(tmp_path / "raw.ply")), gd.read_ply_vertices(str(tmp_path / "clean.ply"))
    assert 0.9 * len(raw) < len(clean) <= len(raw)
    import ref_clean
    pts = np.stack([raw["x"], raw["y"], raw["z"]], 1)
    keep, _ = ref_clean.statistical_outlier_mask(ref_clean.knn_mean_distance(pts), 10.0)
    assert len(clean) == int(keep.sum())
    assert np.array_equal(np.stack([clean["x"], clean["y"], clean["z"]], 1), pts[keep])


def test_grid_build_writes_exactly_cells_plus_two_offsets(emu):
    """g2pc_outlier_grid_build's cell_start holds cells + 2 words (exclusive offsets, the last two = m): found by
    tools/experiments/knn_fuzz.py when the driver still allocated cells + 1 and the range kernel wrote one word past it."""
    import ctypes as C
    import mesh_handler  # noqa: F401  (binds the entry points)
    nv = emu
    L = nv.lib()
    rng = np.random.default_rng(3)
    m = 500
    pts = torch.from_numpy((rng.uniform(-1, 1, size=(m, 3)) * np.array([1.0, 1e-3, 50.0])).astype(np.float32))
    dims_l = [3, 1, 40]
    cells = dims_l[0] * dims_l[1] * dims_l[2]
    guard = 0x7FFFFFFF
    cell_start = torch.full((cells + 2 + 8,), guard, dtype=torch.int32)
    sorted_pos = torch.empty((m, 4), dtype=torch.float32)
    wb = L.g2pc_outlier_grid_workspace(m)
    ws = nv.workspace(wb, "cpu")
    origin = (C.c_float * 3)(-1.0, -1e-3, -50.0)
    dims = (C.c_int32 * 3)(*dims_l)
    nv.check(L.g2pc_outlier_grid_build(nv.ptr(pts), m, C.byref(origin), C.c_float(2.6), C.byref(dims), nv.ptr(sorted_pos),
                                       nv.ptr(cell_start), None, nv.ptr(ws), wb, None), "grid_build")
    cs = cell_start.numpy()
    assert (cs[cells + 2:] == guard).all() and cs[cells] == m and cs[cells + 1] == m and cs[0] == 0
    assert (np.diff(cs[:cells + 1]) >= 0).all()
