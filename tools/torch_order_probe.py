"""Container-only measurement (needs /root/reference): in which ORDER does torch's CPU build evaluate the reference's
per-Gaussian projection arithmetic (gauss_render.py:101-193, 404-437)?

Part 1 (`bisect`): candidate evaluation orders of every matmul of build_covariance_2d / projection_ndc, emulated in numpy
        (fma(a, b, c) = f32(f64(a) * f64(b) + f64(c))), against the UNTOUCHED reference's tensors: number of elements that
        differ in any bit.  The order csrc/py_project.inl implements is the one with 0 everywhere.
Part 2 (`verify`): the library's helper entries (g2pc_projection_ndc, g2pc_build_covariance_2d, g2pc_get_radius -- the
        same device functions k_preprocess_py calls), through the CPU emulator build, against the reference: bit mismatches
        of p_view, p_proj, cov2d, radius, means2D over the whole scene.

Usage: python tools/torch_order_probe.py [bisect|verify] [n] [camera ...]       (defaults: both, 200000, cameras 0 17)
"""
import json
import os
import sys
from math import tan

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "3dgs-to-pc_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import ref_shim                                                     # noqa: E402
from g2pc.synth import make_scene, make_cameras                     # noqa: E402

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def mul(a, b):
    return (np.asarray(a, f32) * np.asarray(b, f32)).astype(f32)


def add(a, b):
    return (np.asarray(a, f32) + np.asarray(b, f32)).astype(f32)


def div(a, b):
    return (np.asarray(a, f32) / np.asarray(b, f32)).astype(f32)


def dot_fma(terms, order=None):
    order = order or range(len(terms))
    terms = [terms[k] for k in order]
    acc = mul(*terms[0])
    for a, b in terms[1:]:
        acc = fma(a, b, acc)
    return acc


def dot_plain(terms, order=None):
    order = order or range(len(terms))
    terms = [terms[k] for k in order]
    acc = mul(*terms[0])
    for a, b in terms[1:]:
        acc = add(acc, mul(a, b))
    return acc


def mism(a, b):
    return int((np.ascontiguousarray(a, f32).view(np.uint32) != np.ascontiguousarray(b, f32).view(np.uint32)).sum())


def bisect(ref, sc, cam, G):
    gr = ref["gauss_render"]
    V, P = cam.world_view_transform.numpy().astype(f32), cam.projection_matrix.numpy().astype(f32)
    po = torch.cat([sc.xyz, torch.ones_like(sc.xyz[:, :1])], dim=-1)
    pv_t = (po @ cam.world_view_transform).numpy()
    ph_t = (po @ cam.world_view_transform @ cam.projection_matrix).numpy()
    A = po.numpy()
    out = {"p_view": {}, "p_hom": {}, "cov2d": {}}
    for name, fn, order in (("fma_forward", dot_fma, None), ("fma_reverse", dot_fma, [3, 2, 1, 0]), ("plain_forward", dot_plain, None),
                            ("plain_reverse", dot_plain, [3, 2, 1, 0])):
        out["p_view"][name] = sum(mism(fn([(A[:, k], V[k, j]) for k in range(4)], order), pv_t[:, j]) for j in range(4))
        out["p_hom"][name] = sum(mism(fn([(pv_t[:, k], P[k, j]) for k in range(4)], order), ph_t[:, j]) for j in range(4))
    with ref_shim.CudaToCpu():
        cov2d_t = gr.build_covariance_2d(G.xyz, G.covariances, cam.world_view_transform, cam.FoVx, cam.FoVy, cam.focal_x,
                                         cam.focal_y).numpy()
    cov3 = G.covariances.numpy()
    x, y, z = [sc.xyz.numpy()[:, i] for i in range(3)]
    t = [add(dot_fma([(x, V[0, j]), (y, V[1, j]), (z, V[2, j])]), V[3, j]) for j in range(3)]
    limx, limy = f32(tan(cam.FoVx * 0.5) * 1.3), f32(tan(cam.FoVy * 0.5) * 1.3)
    tx, ty, tz = mul(np.clip(div(t[0], t[2]), -limx, limx), t[2]), mul(np.clip(div(t[1], t[2]), -limy, limy), t[2]), t[2]
    fx, fy = f32(cam.focal_x), f32(cam.focal_y)
    rz = div(f32(1.0), tz)
    j00, j02, j11, j12 = mul(rz, fx), mul(div(-tx, mul(tz, tz)), fx), mul(rz, fy), mul(div(-ty, mul(tz, tz)), fy)
    zero = np.zeros_like(j00)
    J = [[j00, zero, j02], [zero, j11, j12]]
    W = V[:3, :3].T
    for names in [(a, b, c, d) for a in ("fma", "plain") for b in ("fma", "plain") for c in ("fma", "plain") for d in ("fma", "plain")]:
        d1, d2, d3, d4 = [dot_fma if m == "fma" else dot_plain for m in names]
        M = [[d1([(J[a][k], W[k, c]) for k in range(3)]) for c in range(3)] for a in range(2)]
        A2 = [[d2([(M[a][k], cov3[:, k, c]) for k in range(3)]) for c in range(3)] for a in range(2)]
        B = [[d3([(A2[a][k], V[k, c]) for k in range(3)]) for c in range(3)] for a in range(2)]
        Cm = [[d4([(B[a][k], J[c][k]) for k in range(3)]) for c in range(2)] for a in range(2)]
        out["cov2d"]["JW=%s,@S=%s,@Wt=%s,@Jt=%s" % names] = (
            mism(add(Cm[0][0], f32(0.3)), cov2d_t[:, 0, 0]) + mism(add(Cm[1][1], f32(0.3)), cov2d_t[:, 1, 1]) +
            mism(Cm[0][1], cov2d_t[:, 0, 1]) + mism(Cm[1][0], cov2d_t[:, 1, 0]))
    return out


def verify(ref, sc, cam, G):
    from emu_util import build_emu
    from g2pc import _native as nv
    nv._inject_for_tests(build_emu())
    import gauss_render as mine
    gr = ref["gauss_render"]
    with ref_shim.CudaToCpu():
        cov2d_t = gr.build_covariance_2d(G.xyz, G.covariances, cam.world_view_transform, cam.FoVx, cam.FoVy, cam.focal_x, cam.focal_y)
        ndc_t, view_t, mask_t = gr.projection_ndc(G.xyz, cam.world_view_transform, cam.projection_matrix)
        rad_t = gr.get_radius(cov2d_t)
    cov2d = mine.build_covariance_2d(sc.xyz, G.covariances, cam.world_view_transform, cam.FoVx, cam.FoVy, cam.focal_x, cam.focal_y)
    ndc, view, mask = mine.projection_ndc(sc.xyz, cam.world_view_transform, cam.projection_matrix)
    rad = mine.get_radius(cov2d)
    mx_t = ((ndc_t[..., 0] + 1) * cam.image_width - 1.0) * 0.5
    mx = ((ndc[..., 0] + 1) * cam.image_width - 1.0) * 0.5
    return dict(n=int(sc.xyz.shape[0]), cov2d=mism(cov2d.numpy(), cov2d_t.numpy()), p_proj=mism(ndc.numpy(), ndc_t.numpy()),
                p_view=mism(view.numpy(), view_t.numpy()), in_mask=int((mask != mask_t).sum()),
                radius=mism(rad.numpy(), rad_t.numpy()), means2D_x=mism(mx.numpy(), mx_t.numpy()))


if __name__ == "__main__":
    args = sys.argv[1:]
    what = [a for a in args if a in ("bisect", "verify")] or ["bisect", "verify"]
    nums = [int(a) for a in args if a.isdigit()]
    n = nums[0] if nums else 200_000
    cams = nums[1:] or [0, 17]
    torch.set_num_threads(int(os.environ.get("G2PC_TORCH_THREADS", "8")))
    ref = ref_shim.load_reference()
    sc = make_scene(n, 1237)
    tr, intr = make_cameras(50)
    names = sorted(tr)
    with ref_shim.CudaToCpu():
        G = ref["gauss_handler"].Gaussians(sc.xyz.clone(), sc.scales.clone(), sc.rots.clone(), sc.colours.double(), sc.opacities.clone())
    for ci in cams:
        with ref_shim.CudaToCpu():
            cam = ref["camera_handler"].get_camera("python", torch.tensor(tr[names[ci]]), intr[names[ci]], colour_resolution=1280)
        for w in what:
            print(json.dumps({"camera": ci, "what": w, "mkl_threads": torch.get_num_threads(),
                              "result": (bisect if w == "bisect" else verify)(ref, sc, cam, G)}))
