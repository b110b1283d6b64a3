"""GPU box diagnostic: k_build_cov on the MI355X against the CPU emulator build of the same source, element by element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3dgs-to-pc_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
from g2pc import _native as nv, ops
from g2pc.synth import make_scene
n = 200_000
sc = make_scene(n, 1237, with_sh=False)
dev = torch.device("cuda:0")
cov_g, _, _ = ops.build_covariances(sc.scales.to(dev), sc.rots.to(dev), 1.0)
rot_g = None
cg = cov_g.cpu().numpy().reshape(n, 9)
saved = (nv._LIB, nv._EMULATED)
nv._inject_for_tests(os.path.join(ROOT, "tests", "hipemu", "libg2pc_emu.so"))
cov_e, _, _ = ops.build_covariances(sc.scales, sc.rots, 1.0)
ce = cov_e.numpy().reshape(n, 9)
nv._LIB, nv._EMULATED = saved
d = cg.view(np.uint32) != ce.view(np.uint32)
print("rows differing", int(d.any(axis=1).sum()), "of", n, "per element", d.sum(axis=0).tolist())
# diagonal-only scene: identity rotations -> cov = diag(e^2): isolates exp
rots = torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1)
a = ops.build_covariances(sc.scales.to(dev), rots.to(dev), 1.0)[0].cpu().numpy().reshape(n, 9)
nv._inject_for_tests(os.path.join(ROOT, "tests", "hipemu", "libg2pc_emu.so"))
b = ops.build_covariances(sc.scales, rots, 1.0)[0].numpy().reshape(n, 9)
nv._LIB, nv._EMULATED = saved
print("identity rotations: rows differing", int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum()))
rows = np.nonzero(d.any(axis=1))[0][:3]
for r in rows:
    print("row", r, "q", sc.rots[r].tolist(), "s", sc.scales[r].tolist())
    print("  gpu", cg[r].tolist())
    print("  emu", ce[r].tolist())
