"""CPU (emulator): the exact grid kNN of the outlier removal (csrc/clean.hip, mesh_handler.knn_mean_distance) on random and
degenerate clouds (isotropic, strongly anisotropic extents, many exact duplicates, collinear, tight blob + far floaters, all
identical; k = 1 .. 32) against the KD-tree restatement (oracle/ref_clean.py).  usage: python tools/experiments/knn_fuzz.py
<seed> <cases>.  Round 3: found the cell_start array one word short (heap corruption on the CPU build); 120 cases equal after."""
import os
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3dgs-to-pc_amd'), os.path.join(ROOT, 'oracle')]
import numpy as np, torch
from g2pc import _native as nv
GPU = "gpu" in sys.argv[3:]                   # gpu: the real library on cuda:0 (GPU box), the oracle on the host
if not GPU:
    from emu_util import build_emu
    nv._inject_for_tests(build_emu())
import mesh_handler, ref_clean
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    m = int(rng.integers(1, 4000)); k = int(rng.choice([1, 5, 20, 32])); kind = int(rng.integers(0, 6))
    r = np.random.default_rng(9000 + it)
    if kind == 0: pts = r.normal(size=(m, 3))
    elif kind == 1: pts = r.uniform(-1, 1, size=(m, 3)) * np.array([1.0, 1e-3, 50.0])            # anisotropic extent
    elif kind == 2: pts = np.repeat(r.normal(size=(max(1, m // 7), 3)), 7, axis=0)[:m]             # many exact duplicates
    elif kind == 3: pts = np.stack([np.linspace(0, 1, m), np.zeros(m), np.zeros(m)], 1)            # collinear
    elif kind == 4: pts = np.concatenate([r.normal(size=(m, 3)) * 1e-3, r.normal(size=(max(1, m // 50), 3)) * 100.0])  # tight blob + far floaters
    else: pts = np.zeros((m, 3)) + 0.5                                                             # all identical
    pts = pts.astype(np.float32)
    t = time.time()
    try:
        avg = mesh_handler.knn_mean_distance(torch.from_numpy(pts).to('cuda:0' if GPU else 'cpu'), k).cpu().numpy()
        ref = ref_clean.knn_mean_distance(pts, k)
        ok = avg.shape == ref.shape and np.allclose(avg, ref, rtol=1e-12, atol=1e-300, equal_nan=True)
    except Exception as e:
        ok = False; print(it, m, k, kind, "EXC", type(e).__name__, str(e)[:120])
    bad += (not ok)
    if not ok: print(it, m, k, kind, "MISMATCH")
print("cases", sys.argv[2], "mismatches", bad)
