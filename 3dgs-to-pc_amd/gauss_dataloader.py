"""
Drop-in mirror of the reference's ``gauss_dataloader.py``: ``load_gaussians`` for 3DGS ``.ply``
and ``.splat`` files, ``save_xyz_to_ply`` for the coloured cloud.  No ``plyfile`` dependency (own header parser),
device-agnostic (tensors land on the first GPU when there is one), and the writer packs the binary vertex records on
the GPU (``g2pc_pack_ply_vertices``) and streams them through double-buffered pinned memory instead of building
numpy structured arrays per chunk.  Reference lines: gauss_dataloader.py:8-115 (readers), :118-202 (writer).
"""
import os

import numpy as np
import torch

from g2pc import _native as nv

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def computeColorFromLowDegSH(sh):
    """gauss_dataloader.py:8-14 -- DC coefficient -> RGB in [0, 1] (float64, as the reference)."""
    SH_C0 = 0.28209479177387814
    return ((SH_C0 * sh[:, :, 0].to(torch.double)) + 0.5).clip(0, 1).type(torch.double)


def read_ply_vertices(path):
    """First element of a PLY file (ascii / binary_little_endian / binary_big_endian) as a numpy structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise AttributeError("%s is not a PLY file" % path)
        fmt, count, props, in_first, seen_first = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise AttributeError("PLY header of %s is truncated" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_first = not seen_first
                if in_first:
                    count, seen_first = int(tok[2]), True
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise AttributeError("list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.zeros(count, dtype=[(n, t) for n, t in props])
            for k, (n, _) in enumerate(props):
                out[n] = data[:, k]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dtype = np.dtype([(n, order + t) for n, t in props])
        return np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype, count=count)


def load_ply_data(path, max_sh_degree=3):
    """gauss_dataloader.py:16-88."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    dev = _device()
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    opacities = np.asarray(v["opacity"])[..., np.newaxis]

    if "f_dc_0" in names:
        features_dc = np.zeros((xyz.shape[0], 3, 1))
        for c in range(3):
            features_dc[:, c, 0] = v["f_dc_%d" % c]
        extra = sorted([n for n in names if n.startswith("f_rest_")], key=lambda x: int(x.split('_')[-1]))
        assert len(extra) == 3 * (max_sh_degree + 1) ** 2 - 3
        features_extra = np.zeros((xyz.shape[0], len(extra)))
        for idx, n in enumerate(extra):
            features_extra[:, idx] = v[n]
        features_extra = features_extra.reshape((features_extra.shape[0], 3, (max_sh_degree + 1) ** 2 - 1))
        features_all = torch.cat((torch.tensor(features_dc, device=dev), torch.tensor(features_extra, device=dev)), 2)
        colours = computeColorFromLowDegSH(features_all)
    elif "red" in names:
        colours = torch.zeros((xyz.shape[0], 3), device=dev, dtype=torch.double)
        for c, n in enumerate(("red", "green", "blue")):
            colours[:, c] = torch.tensor(np.asarray(v[n]), device=dev, dtype=torch.double)
        if torch.count_nonzero(colours > 1.0) > 0:
            colours /= 255
            colours = colours.clip(0, 1)
        features_all = None
    else:
        raise AttributeError("Input ply file does not have valid colours (must have either spherical harmoics or RGB colour fields)")

    scale_names = sorted([n for n in names if n.startswith("scale_")], key=lambda x: int(x.split('_')[-1]))
    scales = np.stack([v[n] for n in scale_names], axis=1).astype(np.float64)
    rot_names = sorted([n for n in names if n.startswith("rot")], key=lambda x: int(x.split('_')[-1]))
    rots = np.stack([v[n] for n in rot_names], axis=1).astype(np.float64)

    opacities = (1 / (1 + torch.exp(torch.tensor(-opacities, device=dev)))).type(torch.float).squeeze(1)
    xyz_tensor = torch.tensor(xyz, device=dev)
    scales_tensor = torch.tensor(scales, device=dev)
    rots_tensor = torch.tensor(rots / np.expand_dims(np.linalg.norm(rots, axis=1), 1), device=dev)
    return xyz_tensor, scales_tensor, rots_tensor, colours, opacities, features_all


def load_splat_data(path):
    """gauss_dataloader.py:90-115 -- 32-byte records: xyz f32x3, scales f32x3, rgba u8x4, rotation u8x4."""
    with open(path, "rb") as f:
        content = f.read()
    dtype = np.dtype([('xyz', np.float32, 3), ('scales', np.float32, 3), ('colour', np.uint8, 4), ('rots', np.uint8, 4)])
    data = np.frombuffer(content, dtype=dtype, count=len(content) // dtype.itemsize)
    dev = _device()
    xyz_tensor = torch.tensor(data['xyz'], device=dev)
    scales_tensor = torch.tensor(np.log(data['scales']), device=dev)
    colours_tensor = torch.tensor(data['colour'][:, :3] / 255, device=dev)
    opacities = torch.tensor(data['colour'][:, 3] / 255, device=dev)
    rots_tensor = torch.tensor((data['rots'].astype(np.float32) - 128) / 128, device=dev)
    return xyz_tensor, scales_tensor, rots_tensor, colours_tensor, opacities, None


def save_xyz_to_ply(xyz_points, filename, rgb_colors=None, normals_points=None, chunk_size=10 ** 6, quiet=False):
    """gauss_dataloader.py:118-202 -- same header, same 15/27-byte records, same uint8 truncation of the colours."""
    assert xyz_points.shape[1] == 3, "Input points should be in the format (N, 3)"
    total = xyz_points.shape[0]
    if rgb_colors is None:
        rgb_colors = torch.full((total, 3), 255.0, device=xyz_points.device)
    props = "property float x\nproperty float y\nproperty float z\n"
    if normals_points is not None:
        props += "property float nx\nproperty float ny\nproperty float nz\n"
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n%sproperty uchar red\nproperty uchar green\n"
              "property uchar blue\nend_header\n" % (total, props))
    rec = 27 if normals_points is not None else 15
    on_gpu = xyz_points.device.type == "cuda" or nv.emulated()
    with open(filename, "wb") as ply_file:
        ply_file.write(header.encode("utf-8"))
        if total == 0:
            return
        if not on_gpu:
            raise nv.G2pcError("save_xyz_to_ply packs the records with a HIP kernel: the cloud must live in HBM")
        pts = xyz_points.to(torch.float32).contiguous()
        cols = rgb_colors.to(torch.float32).contiguous()
        nrm = normals_points.to(torch.float32).contiguous() if normals_points is not None else None
        dev = pts.device
        pad = lambda m: (m * rec + 3) // 4 * 4 + 256 * rec
        staging = [torch.empty((pad(min(chunk_size, total)),), dtype=torch.uint8, device=dev) for _ in range(2)]
        pinned = [torch.empty((pad(min(chunk_size, total)),), dtype=torch.uint8) for _ in range(2)]
        if dev.type == "cuda":
            pinned = [p.pin_memory() for p in pinned]
        events = [None, None]
        chunks = [(s, min(s + chunk_size, total)) for s in range(0, total, chunk_size)]

        def launch(i):
            s, e = chunks[i]
            b = i % 2
            nv.check(nv.lib().g2pc_pack_ply_vertices(nv.ptr(pts[s:e]), nv.ptr(nrm[s:e]) if nrm is not None else None,
                                                     nv.ptr(cols[s:e]), e - s, nv.ptr(staging[b]), nv.stream_handle(dev)),
                     "pack_ply_vertices")
            pinned[b][:(e - s) * rec].copy_(staging[b][:(e - s) * rec], non_blocking=True)
            if dev.type == "cuda":
                events[b] = torch.cuda.Event()
                events[b].record()

        launch(0)
        for i, (s, e) in enumerate(chunks):
            if i + 1 < len(chunks):
                launch(i + 1)                                  # pack + copy of the next chunk overlap this chunk's write
            if events[i % 2] is not None:
                events[i % 2].synchronize()
            ply_file.write(pinned[i % 2][:(e - s) * rec].numpy().tobytes())


def load_gaussians(input_path, max_sh_degree=3):
    """gauss_dataloader.py:204-211."""
    ext = os.path.splitext(input_path)[1]
    if ext == ".splat":
        return load_splat_data(input_path)
    elif ext == ".ply":
        return load_ply_data(input_path, max_sh_degree=max_sh_degree)
    raise AttributeError(f"Unsupported input type {ext}")
