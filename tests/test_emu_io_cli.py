"""The I/O layer around the hot path ("next" rows f1/f2) and the reference's CLI end to end, on CPU through the emulator:
3DGS .ply / .splat readers, transforms.json / COLMAP readers (against the reference's own parser when it is present),
the GPU-packed PLY writer, and `gauss_to_pc.main([...])`."""
import json
import os
import struct
import sys

import numpy as np
import pytest
import torch

from emu_util import emu  # noqa: F401
from g2pc.synth import make_scene, make_cameras


def _write_3dgs_ply(path, sc, degree=3):
    n = sc.xyz.shape[0]
    k = (degree + 1) ** 2
    names = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + \
            ["f_rest_%d" % i for i in range(3 * k - 3)] + ["opacity"] + ["scale_%d" % i for i in range(3)] + \
            ["rot_%d" % i for i in range(4)]
    arr = np.zeros(n, dtype=[(nm, "<f4") for nm in names])
    arr["x"], arr["y"], arr["z"] = sc.xyz[:, 0].numpy(), sc.xyz[:, 1].numpy(), sc.xyz[:, 2].numpy()
    dc = (sc.colours.numpy() - 0.5) / 0.28209479177387814
    for i in range(3):
        arr["f_dc_%d" % i] = dc[:, i]
        arr["scale_%d" % i] = sc.scales[:, i].numpy()
    op = sc.opacities.numpy().clip(1e-4, 1 - 1e-4)
    arr["opacity"] = np.log(op / (1 - op))
    for i in range(4):
        arr["rot_%d" % i] = sc.rots[:, i].numpy()
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
        for nm in names:
            f.write(("property float %s\n" % nm).encode())
        f.write(b"end_header\n")
        f.write(arr.tobytes())


def test_ply_and_splat_readers(emu, tmp_path):
    import gauss_dataloader as gd
    sc = make_scene(500, 3)
    _write_3dgs_ply(tmp_path / "scene.ply", sc)
    xyz, scales, rots, colours, opac, shs = gd.load_gaussians(str(tmp_path / "scene.ply"))
    assert torch.allclose(xyz.float(), sc.xyz) and torch.allclose(scales.float(), sc.scales)
    assert torch.allclose(rots.float(), sc.rots, atol=1e-6)
    inner = (sc.colours > 1e-3).all(1) & (sc.colours < 1 - 1e-3).all(1)
    assert torch.allclose(colours.float()[inner], sc.colours[inner], atol=1e-6) and colours.dtype == torch.double
    assert torch.allclose(opac, sc.opacities.clip(1e-4, 1 - 1e-4), atol=1e-6) and shs.shape == (500, 3, 16)
    rec = np.zeros(7, dtype=[('xyz', np.float32, 3), ('scales', np.float32, 3), ('colour', np.uint8, 4), ('rots', np.uint8, 4)])
    rec['xyz'] = np.arange(21).reshape(7, 3); rec['scales'] = 0.5; rec['colour'] = [255, 0, 128, 64]; rec['rots'] = [255, 128, 128, 128]
    (tmp_path / "s.splat").write_bytes(rec.tobytes())
    xyz, scales, rots, colours, opac, shs = gd.load_gaussians(str(tmp_path / "s.splat"))
    assert xyz.shape == (7, 3) and torch.allclose(scales, torch.full((7, 3), float(np.log(0.5))))
    assert torch.allclose(colours[0].float(), torch.tensor([1.0, 0.0, 128 / 255])) and abs(float(opac[0]) - 64 / 255) < 1e-7
    assert shs is None and abs(float(rots[0, 0]) - 127 / 128) < 1e-7


def test_ply_writer_matches_numpy_packing(emu, tmp_path):
    import gauss_dataloader as gd
    g = torch.Generator().manual_seed(1)
    m = 1000
    pts, nrm = torch.randn((m, 3), generator=g), torch.randn((m, 3), generator=g)
    cols = torch.rand((m, 3), generator=g) * 255
    for normals, rec in ((nrm, 27), (None, 15)):
        out = tmp_path / ("cloud%d.ply" % rec)
        gd.save_xyz_to_ply(pts, str(out), rgb_colors=cols, normals_points=normals, chunk_size=300, quiet=True)
        v = gd.read_ply_vertices(str(out))
        assert len(v) == m and v.dtype.itemsize == rec
        assert np.array_equal(np.stack([v["x"], v["y"], v["z"]], 1), pts.numpy())
        assert np.array_equal(np.stack([v["red"], v["green"], v["blue"]], 1), cols.numpy().astype(np.uint8))   # truncation
        if normals is not None:
            assert np.array_equal(np.stack([v["nx"], v["ny"], v["nz"]], 1), nrm.numpy())


def _ref_transform_loader():
    try:
        from ref_shim import load_reference, reference_available
        if not reference_available():
            return None
        load_reference()
        sys.path.insert(0, "/root/reference")
        saved = sys.modules.pop("transform_dataloader", None)
        try:
            import importlib
            mod = importlib.import_module("transform_dataloader")
        finally:
            sys.modules.pop("transform_dataloader", None)
            if saved is not None:
                sys.modules["transform_dataloader"] = saved
            sys.path.remove("/root/reference")
        return mod
    except Exception:
        return None


def test_transform_readers(emu, tmp_path):
    import transform_dataloader as td
    tr, intr = make_cameras(5, width=640, height=360, focal=500.0)
    frames = [{"file_path": "images/%s.png" % k, "transform_matrix": tr[k]} for k in tr]
    (tmp_path / "transforms.json").write_text(json.dumps({"w": 640, "h": 360, "fl_x": 500.0, "frames": frames}))
    t2, i2 = td.load_transform_data(str(tmp_path / "transforms.json"), skip_rate=1)
    assert list(t2) == ["cam_0000", "cam_0002", "cam_0004"] and i2["cam_0000"] == [640, 360, 500.0, 500.0]
    assert t2["cam_0002"] == tr["cam_0002"]
    # COLMAP text + binary with the same content
    col = tmp_path / "colmap"
    col.mkdir()
    (col / "cameras.txt").write_text("# cams\n1 PINHOLE 640 360 500.0 510.0 320.0 180.0\n")
    lines, recs = ["# images"], []
    rng = np.random.default_rng(0)
    for i in range(4):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(size=3)
        lines += ["%d %s 1 img_%d.jpg" % (i + 1, " ".join(repr(float(v)) for v in list(q) + list(t)), i), "0.0 0.0 -1"]
        recs.append((i + 1, q, t, "img_%d.jpg" % i))
    (col / "images.txt").write_text("\n".join(lines) + "\n")
    t_txt, i_txt = td.load_transform_data(str(col))
    binp = tmp_path / "colmap_bin" / "sparse" / "0"
    binp.mkdir(parents=True)
    with open(binp / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", 1) + struct.pack("<iiQQdddd", 1, 1, 640, 360, 500.0, 510.0, 320.0, 180.0))
    with open(binp / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", len(recs)))
        for iid, q, t, name in recs:
            f.write(struct.pack("<idddddddi", iid, *q, *t, 1) + name.encode() + b"\x00" + struct.pack("<Q", 2))
            f.write(struct.pack("<ddqddq", 0.0, 0.0, -1, 1.0, 1.0, -1))
    t_bin, i_bin = td.load_transform_data(str(tmp_path / "colmap_bin"))
    assert list(t_txt) == list(t_bin) == ["img_%d" % i for i in range(4)]
    for k in t_txt:
        np.testing.assert_allclose(np.array(t_txt[k]), np.array(t_bin[k]), atol=1e-12)
        c2w = np.array(t_txt[k])
        np.testing.assert_allclose(c2w[:3, :3] @ c2w[:3, :3].T, np.eye(3), atol=1e-9)       # a rigid pose
    assert i_bin["img_0"][:4] == (640, 360, 500.0, 510.0) and i_txt["img_0"][:4] == ["640", "360", "500.0", "510.0"]
    ref = _ref_transform_loader()
    if ref is not None:                     # authoring container: the reference's own parser must agree
        for path, kw in ((str(tmp_path / "transforms.json"), dict(skip_rate=1)), (str(col), {}), (str(tmp_path / "colmap_bin"), {})):
            a, b = ref.load_transform_data(path, **kw), td.load_transform_data(path, **kw)
            assert list(a[0]) == list(b[0]) and a[1] == b[1]
            for k in a[0]:
                assert np.array_equal(np.array(a[0][k]), np.array(b[0][k]))


def test_cli_end_to_end(emu, tmp_path, capsys):
    """python gauss_to_pc.py --input_path scene.ply --transform_path transforms.json --renderer_type python ..."""
    import gauss_to_pc as g2p
    import gauss_dataloader as gd
    sc = make_scene(1500, 8, scale_lo=0.01, scale_hi=0.06)
    _write_3dgs_ply(tmp_path / "scene.ply", sc)
    tr, intr = make_cameras(2, width=180, height=101, focal=155.0)
    frames = [{"file_path": "%s.png" % k, "transform_matrix": tr[k]} for k in tr]
    (tmp_path / "transforms.json").write_text(json.dumps({"w": 180, "h": 101, "fl_x": 155.0, "frames": frames}))
    out = tmp_path / "pc.ply"
    g2p.main(["--input_path", str(tmp_path / "scene.ply"), "--transform_path", str(tmp_path / "transforms.json"),
              "--renderer_type", "python", "--num_points", "20000", "--colour_quality", "original",
              "--output_path", str(out), "--quiet"])
    v = gd.read_ply_vertices(str(out))
    assert abs(len(v) - 20000) < 400 and v.dtype.itemsize == 27
    assert np.isfinite(v["x"]).all() and v["red"].max() > 50
    with pytest.raises(AttributeError):
        g2p.config_parser(["--input_path", "x.ply"])                      # transforms required unless --no_render_colours
    with pytest.raises(AttributeError):
        g2p.config_parser(["--input_path", "x.ply", "--no_render_colours", "--min_opacity", "2"])
