"""
Seeded synthetic 3DGS scenes and camera rigs.

There is no network and the reference ships no sample scene, so every test / bench input is
generated here.  Everything is produced by a CPU ``torch.Generator`` so a given
``(n, seed)`` is bit-identical on the authoring container and on the GPU box.

Scene (all float32):
    xyz        ~ U(-1, 1)^3
    log-scales = log(U(0.002, 0.02)) per axis   (reference stores log-space scales,
                                                 gauss_handler.py:53-55); evaluated in float64 and
                                                 rounded once, so that hosts agree bit for bit
    rotation   = normalize(N(0, I4)), (r, x, y, z) order (gauss_handler.py:32-35)
    opacity    ~ U(0.05, 1), already activated
    sh_dc      ~ N(0, 1); colour = clip(0.28209479 * dc + 0.5, 0, 1)
                 (DC-only colour, gauss_dataloader.py:63-68 semantics)
    sh_rest    ~ N(0, 0.1), K = 16 coefficients, layout [N, 16, 3] (forward.cu:31)

Cameras: C poses on a Fibonacci sphere of radius 3.5 looking at the origin, up = +y,
NeRF / OpenGL camera-to-world matrices (what transforms.json carries, i.e. the camera
looks down its own -z axis), intrinsics [w, h, fl_x, fl_y] = [1280, 720, 1100, 1100].
"""
from __future__ import annotations

import math
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np
import torch

SH_C0 = 0.28209479177387814


class SyntheticScene(NamedTuple):
    xyz: torch.Tensor        # [N,3] f32
    scales: torch.Tensor     # [N,3] f32, log-space
    rots: torch.Tensor       # [N,4] f32, unit quaternion (r,x,y,z)
    opacities: torch.Tensor  # [N]   f32
    colours: torch.Tensor    # [N,3] f32 in [0,1]
    shs: Optional[torch.Tensor]  # [N,16,3] f32 or None


def make_scene(n: int, seed: int = 1234, with_sh: bool = False, device="cpu",
               scale_lo: float = 0.002, scale_hi: float = 0.02) -> SyntheticScene:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    xyz = torch.rand((n, 3), generator=g, dtype=torch.float32) * 2.0 - 1.0
    s = torch.rand((n, 3), generator=g, dtype=torch.float32) * (scale_hi - scale_lo) + scale_lo
    # log in float64, rounded once: the same bits on every host.  torch.log on float32 is a vendor vector routine -- on the
    # authoring container's Xeon it returns the correctly rounded value for all but 3e-5 of its arguments, on the GPU box's
    # EPYC for all but 1.6 % (tools/experiments/scene_hash.py): until round 4 every fixture was compared on the GPU box with
    # a scene whose log-scales differed from the generator's in 1.6 % of the elements (7.5 % of the covariance rows)
    scales = torch.from_numpy(np.log(s.numpy().astype(np.float64)).astype(np.float32))
    q = torch.randn((n, 4), generator=g, dtype=torch.float32)
    rots = q / q.norm(dim=1, keepdim=True)
    opac = torch.rand((n,), generator=g, dtype=torch.float32) * 0.95 + 0.05
    dc = torch.randn((n, 3), generator=g, dtype=torch.float32)
    colours = (SH_C0 * dc + 0.5).clamp(0.0, 1.0)
    shs = None
    if with_sh:
        rest = torch.randn((n, 15, 3), generator=g, dtype=torch.float32) * 0.1
        shs = torch.cat([dc[:, None, :], rest], dim=1).contiguous()
    out = SyntheticScene(xyz, scales, rots, opac, colours, shs)
    if str(device) != "cpu":
        out = SyntheticScene(*[t.to(device) if t is not None else None for t in out])
    return out


def _look_at_c2w(eye: torch.Tensor) -> torch.Tensor:
    """OpenGL camera-to-world: camera looks down -z towards the origin, +y up."""
    fwd = -eye / eye.norm()                      # viewing direction
    up = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    if abs(float(fwd @ up)) > 0.999:             # pole: pick another up
        up = torch.tensor([1.0, 0.0, 0.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0] = right
    c2w[:3, 1] = true_up
    c2w[:3, 2] = -fwd
    c2w[:3, 3] = eye
    return c2w


def make_cameras(c: int, radius: float = 3.5, width: int = 1280, height: int = 720,
                 focal: float = 1100.0) -> Tuple[Dict[str, List[List[float]]], Dict[str, List[float]]]:
    """Returns (transforms, intrinsics) shaped like transform_dataloader.load_transform_data:
    name -> 4x4 nested list (c2w), name -> [w, h, fl_x, fl_y]."""
    golden = math.pi * (3.0 - math.sqrt(5.0))
    transforms, intrinsics = {}, {}
    for i in range(c):
        y = 1.0 - 2.0 * (i + 0.5) / c
        r = math.sqrt(max(0.0, 1.0 - y * y))
        th = golden * i
        eye = torch.tensor([math.cos(th) * r, y, math.sin(th) * r], dtype=torch.float64) * radius
        c2w = _look_at_c2w(eye).to(torch.float32)
        name = "cam_%04d" % i
        transforms[name] = c2w.tolist()
        intrinsics[name] = [float(width), float(height), float(focal), float(focal)]
    return transforms, intrinsics
