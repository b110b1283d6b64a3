"""CPU (emulator): random point-count distributions (log-normal, uniform, constant, sparse spikes, Pareto; binned and exact)
through g2pc_sampler_bin_table against the host bin table (ops.bin_table_from_hist, pinned to the reference by the sampler
fixtures).  usage: python tools/experiments/bin_table_fuzz.py <seed> <cases>.  Round 3: 600 cases, no difference; a constant
distribution makes the reference's np.gradient raise and the device plan report error 2 (ops raises ValueError there too)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3dgs-to-pc_amd'), os.path.join(ROOT, 'oracle')]
import numpy as np, torch
from g2pc import _native as nv
from emu_util import build_emu
nv._inject_for_tests(build_emu())
from g2pc import ops
emu = nv
L = emu.lib(); HL = ops.HIST_GUESS
rng0 = np.random.default_rng(int(sys.argv[1]))
bad = 0
for it in range(int(sys.argv[2])):
    g = int(rng0.integers(1, 30000)); exact = bool(rng0.integers(0, 2))
    kind = int(rng0.integers(0, 5))
    rng = np.random.default_rng(1000 + it)
    if kind == 0: ppg = np.floor(rng.lognormal(np.log(float(rng0.choice([2., 30., 400., 3000.]))), float(rng0.choice([0.3, 0.9, 2.0])), g))
    elif kind == 1: ppg = rng.integers(0, int(rng0.choice([2, 3, 50, 101, 8000])), g).astype(np.float64)
    elif kind == 2: ppg = np.full(g, float(rng0.integers(0, 300)))
    elif kind == 3:
        ppg = np.zeros(g); ppg[rng.integers(0, g, max(1, g // 50))] = float(rng0.integers(1, 8000))
    else:
        ppg = np.floor(rng.pareto(1.2, g) * float(rng0.choice([1., 20.])))
    ppg = np.minimum(ppg, HL - 1).astype(np.int32)
    t = torch.from_numpy(ppg)
    hist_dev = ops.bincount(t, HL)
    stats = torch.tensor([int(ppg.sum()), int((ppg == 0).sum()), 0, int(ppg.max())], dtype=torch.int64)
    lut, quota, bin_lo = (torch.empty((HL,), dtype=torch.int32) for _ in range(3))
    bin_start = torch.empty((HL + 2,), dtype=torch.int32)
    plan = torch.zeros((10,), dtype=torch.int64)
    wb = L.g2pc_sampler_bin_table_workspace(HL); ws = emu.workspace(wb, "cpu")
    rc = L.g2pc_sampler_bin_table(emu.ptr(hist_dev), HL, emu.ptr(stats), int(exact), 1, ops.WAVE_MODE_MIN_DRAWS, emu.ptr(lut), emu.ptr(quota), emu.ptr(bin_start), emu.ptr(bin_lo), ops.C_void(plan), emu.ptr(ws), wb, None)
    B, gv, p_wave, any_s, means_rows, rows_ub, err = [int(v) for v in plan[:7]]
    hist = np.bincount(ppg, minlength=int(ppg.max()) + 1).astype(np.int64)
    try:
        ref = ops.bin_table_from_hist(hist, exact)
    except Exception as e:
        print(it, kind, g, exact, "host raised", type(e).__name__, str(e)[:60], "device rc", rc, "err", err); continue
    ok = rc == 0 and err == 0 and B == len(ref)
    if ok:
        got = list(ops._LazyBins(bin_lo, quota, B))
        ok = got == [(float(s), float(e), int(n)) for s, e, n in ref]
    if ok:
        rl = np.full((int(ppg.max()) + 1,), -1, dtype=np.int32); members = np.zeros((B,), dtype=np.int64)
        for b, (s, e, n) in enumerate(ref):
            lo, hi = int(np.ceil(s)), min(int(np.ceil(e)), int(ppg.max()) + 1)
            if n > 0 and hi > lo:
                rl[lo:hi] = b; members[b] = hist[lo:hi].sum()
        bs = np.concatenate([[0], np.cumsum(members)])
        ok = np.array_equal(lut.numpy()[:rl.shape[0]], rl) and (lut.numpy()[rl.shape[0]:] == -1).all() and np.array_equal(bin_start.numpy()[:B + 1].astype(np.int64), bs) and gv == int(bs[-1])
    if not ok:
        bad += 1; print(it, kind, g, exact, "max", int(ppg.max()), "MISMATCH rc", rc, "err", err, "B", B, len(ref))
print("cases", int(sys.argv[2]), "mismatches", bad)
