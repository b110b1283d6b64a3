"""-m gpu: PLY writer (GPU-packed records, pinned double-buffered read-back) and the CLI on the MI355X."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_ply_writer_large(tmp_path):
    import gauss_dataloader as gd
    g = torch.Generator().manual_seed(2)
    m = 2_500_123
    pts, nrm = torch.randn((m, 3), generator=g), torch.randn((m, 3), generator=g)
    cols = torch.rand((m, 3), generator=g) * 255
    out = tmp_path / "cloud.ply"
    gd.save_xyz_to_ply(pts.to(DEV), str(out), rgb_colors=cols.to(DEV), normals_points=nrm.to(DEV), quiet=True)
    v = gd.read_ply_vertices(str(out))
    assert len(v) == m
    assert np.array_equal(np.stack([v["x"], v["y"], v["z"]], 1), pts.numpy())
    assert np.array_equal(np.stack([v["nx"], v["ny"], v["nz"]], 1), nrm.numpy())
    assert np.array_equal(np.stack([v["red"], v["green"], v["blue"]], 1), cols.numpy().astype(np.uint8))


def test_cli_end_to_end_gpu(tmp_path):
    from test_emu_io_cli import _write_3dgs_ply
    from g2pc.synth import make_scene, make_cameras
    import gauss_to_pc as g2p
    import gauss_dataloader as gd
    sc = make_scene(50_000, 8)
    _write_3dgs_ply(tmp_path / "scene.ply", sc)
    tr, intr = make_cameras(4)
    frames = [{"file_path": "%s.png" % k, "transform_matrix": tr[k]} for k in tr]
    (tmp_path / "transforms.json").write_text(json.dumps({"w": 1280, "h": 720, "fl_x": 1100.0, "frames": frames}))
    for renderer in ("python", "cuda"):
        out = tmp_path / ("pc_%s.ply" % renderer)
        g2p.main(["--input_path", str(tmp_path / "scene.ply"), "--transform_path", str(tmp_path / "transforms.json"),
                  "--renderer_type", renderer, "--num_points", "500000", "--output_path", str(out), "--quiet"])
        v = gd.read_ply_vertices(str(out))
        assert abs(len(v) - 500000) < 5000 and np.isfinite(v["x"]).all()
