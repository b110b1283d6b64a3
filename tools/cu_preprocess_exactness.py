"""Container-only measurement (needs /root/reference): K1 of the native-rasteriser path -- radii, tile rectangles, projected
means, depths, conics -- of the reference's own preprocessCUDA (oracle/_ref, host build with -ffp-contract=off; pass
`fma` as third argument for the contracted build) against k_preprocess_cu (CPU emulator build of csrc/raster.hip: the same
source the GPU runs, contraction off) on a scene large enough to see a 1e-6 mismatch rate.
Usage: python tools/cu_preprocess_exactness.py [n] [ncam] [fma]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "3dgs-to-pc_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import ref_shim                                                     # noqa: E402
import make_golden_cu as MG                                         # noqa: E402
from emu_util import build_emu                                      # noqa: E402
from g2pc import _native as nv                                      # noqa: E402
from g2pc.synth import make_scene, make_cameras                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 2
case = dict(n=n, seed=4242, width=1280, height=720, focal=1100.0, ncam=ncam, scale=(0.002, 0.02), with_sh=False, surf=False,
            mask="none", pixel_stride=64, store_list=False)
ref = ref_shim.load_reference()
cams, _, _ = MG.run_case(ref, case, "synced", len(sys.argv) > 3 and sys.argv[3] == "fma")

nv._inject_for_tests(build_emu())
import camera_handler                                               # noqa: E402
import gauss_render                                                 # noqa: E402
from gauss_handler import Gaussians                                 # noqa: E402
sc = make_scene(n, case["seed"], scale_lo=0.002, scale_hi=0.02)
tr, intr = make_cameras(ncam, width=1280, height=720, focal=1100.0)
G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
R = gauss_render.get_renderer("cuda", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
out = []
for i, nm in enumerate(tr):
    rs = camera_handler.get_camera("cuda", torch.tensor(tr[nm]), intr[nm])
    cam, campos, mask = R._camera(rs)
    R._front(R._sync, cam, campos, rs.sh_degree)
    rad = R._sync.radii.numpy()
    rec = R._sync.rec.numpy().reshape(n, 16)
    rc = R._sync.rect.numpy().astype(np.int64)
    touched = np.where(rad > 0, (((rc >> 8) & 255) - (rc & 255) + 1) * (((rc >> 24) & 255) - ((rc >> 16) & 255) + 1), 0)
    d = cams[i]
    vis = d["radii"] > 0
    m2 = d["means2D"].reshape(n, 2)
    rep = dict(camera=i, visible=int(vis.sum()), radii_mismatch=int((rad != d["radii"]).sum()),
               tiles_touched_mismatch=int((touched != d["tiles_touched"]).sum()),
               num_rendered=(int(R._sync.offsets[n]), d["num_rendered"]),
               means2D_bit_mismatch=int((rec[vis, 0:2].view(np.uint32) != m2[vis].view(np.uint32)).any(axis=1).sum()),
               depth_bit_mismatch=int((rec[vis, 6].view(np.uint32) != d["depths"][vis].view(np.uint32)).sum()))
    # conic: rec holds (-0.5*log2e)*conic.x, (-log2e)*conic.y, (-0.5*log2e)*conic.z: apply the same scaling to the reference's
    co = d["conic_opacity"].reshape(n, 4)
    L2E = np.float32(1.4426950408889634)
    want = np.stack([(np.float32(-0.5) * L2E) * co[:, 0], (-L2E) * co[:, 1], (np.float32(-0.5) * L2E) * co[:, 2]], axis=1).astype(np.float32)
    rep["conic_bit_mismatch"] = int((rec[vis, 2:5].view(np.uint32) != want[vis].view(np.uint32)).any(axis=1).sum())
    rel = np.abs(rec[vis, 2:5] - want[vis]) / np.maximum(np.abs(want[vis]), 1e-30)
    rep["conic_rel_max"] = float(rel.max())
    out.append(rep)
    print(json.dumps(rep))
    if os.environ.get("G2PC_DEBUG_GEOM"):
        a, b = rec[vis, 0:2], m2[vis]
        for ax in (0, 1):
            ne = a[:, ax].view(np.uint32) != b[:, ax].view(np.uint32)
            ulp = np.abs(a[ne, ax].view(np.int32).astype(np.int64) - b[ne, ax].view(np.int32).astype(np.int64))
            print("axis", ax, "mismatch", int(ne.sum()), "ulp hist", np.bincount(np.minimum(ulp, 5)))
        cr = rel.max(axis=1)
        print("conic rel quantiles", np.quantile(cr, [0.5, 0.9, 0.99, 0.999, 1.0]), "bit-equal-ish (<2e-7):", float((cr < 2e-7).mean()))
