"""Sampler kernels (plan / count / emit) through the fiber emulator against the reference's golden outputs."""
import numpy as np
import pytest
import torch

from emu_util import emu  # noqa: F401
from sampler_checks import run_sampler_case, assert_sampler_matches
from g2pc import ops


@pytest.mark.parametrize("name", ["sampler_binned_n3000.npz", "sampler_exact_n3000.npz"])
def test_sampler_matches_reference(emu, golden_dir, name):
    g, sc, ppg, out = run_sampler_case(golden_dir, name)
    assert_sampler_matches(g, ppg, out)
    # every point is attributed to a Gaussian whose quota allows it
    gi = out.gauss_index.numpy()
    counts = np.bincount(gi, minlength=int(g["n"]))
    assert counts.max() <= max(q for _, _, q in out.bins)


def test_sampler_wave_mode_equals_thread_mode(emu, golden_dir, monkeypatch):
    g, sc, ppg, out = run_sampler_case(golden_dir, "sampler_binned_n3000.npz")
    monkeypatch.setattr(ops, "WAVE_MODE_MIN_DRAWS", 4)            # force most bins through the wave kernels
    g2, sc2, ppg2, out2 = run_sampler_case(golden_dir, "sampler_binned_n3000.npz")
    assert torch.equal(out.points, out2.points) and torch.equal(out.gauss_index, out2.gauss_index)
    monkeypatch.setattr(ops, "WAVE_MODE_MIN_DRAWS", 10 ** 9)      # and none
    g3, sc3, ppg3, out3 = run_sampler_case(golden_dir, "sampler_binned_n3000.npz")
    assert torch.equal(out.points, out3.points)


@pytest.mark.parametrize("name", ["sampler_binned_n3000.npz", "sampler_exact_n3000.npz"])
@pytest.mark.parametrize("wave_min", [32, 4])
def test_draw_once_equals_two_pass_sampling(emu, golden_dir, monkeypatch, name, wave_min):
    """The count pass that keeps its points + the copying emission (G2pcSampleStage, the default) against the two-pass form that
    evaluates every keyed draw twice: the same cloud bit for bit, lane-mode and wave-mode staging, 5 and 100 attempts."""
    monkeypatch.setattr(ops, "WAVE_MODE_MIN_DRAWS", wave_min)
    monkeypatch.setattr(ops, "DRAW_ONCE", True)
    monkeypatch.setattr(ops, "ONE_CALL_TAIL", False)
    _, _, _, a = run_sampler_case(golden_dir, name)
    monkeypatch.setattr(ops, "DRAW_ONCE", False)
    _, _, _, b = run_sampler_case(golden_dir, name)
    assert torch.equal(a.points, b.points) and torch.equal(a.colours, b.colours) and torch.equal(a.gauss_index, b.gauss_index)
    # ... and the whole tail as ONE library call over one workspace (g2pc_sampler_run; the binned default's 5 attempts take it)
    monkeypatch.setattr(ops, "DRAW_ONCE", True)
    monkeypatch.setattr(ops, "ONE_CALL_TAIL", True)
    _, _, _, c = run_sampler_case(golden_dir, name)
    assert torch.equal(a.points, c.points) and torch.equal(a.colours, c.colours) and torch.equal(a.gauss_index, c.gauss_index)
    assert list(a.emitted_per_attempt) == list(c.emitted_per_attempt)


@pytest.mark.parametrize("seed,g,scale,exact", [(1, 5000, 30.0, False), (2, 20000, 300.0, False), (3, 3000, 3.0, False),
                                                (4, 4000, 50.0, True), (5, 150, 800.0, False), (6, 9000, 1500.0, False)])
def test_device_bin_table_equals_the_host_table(emu, seed, g, scale, exact):
    """g2pc_sampler_bin_table (the bin heuristics of gauss_to_pc.py:105-138 / :308-337 in one block on the device) against
    the numpy restatement the host path uses (pinned to the reference by the sampler fixtures): the same bins, quotas,
    look-up table, member offsets and plan numbers for point-count distributions of several shapes, binned and exact."""
    import numpy as np
    import torch
    from g2pc import ops
    rng = np.random.default_rng(seed)
    ppg = np.minimum(np.floor(rng.lognormal(np.log(scale), 0.9, g)), ops.HIST_GUESS - 1).astype(np.int32)
    ppg[rng.random(g) < 0.1] = 0
    t = torch.from_numpy(ppg)
    L = emu.lib()
    HL = ops.HIST_GUESS
    hist_dev = ops.bincount(t, HL)
    stats = torch.tensor([int(ppg.sum()), int((ppg == 0).sum()), 0, int(ppg.max())], dtype=torch.int64)
    lut, quota, bin_lo = (torch.empty((HL,), dtype=torch.int32) for _ in range(3))
    bin_start = torch.empty((HL + 2,), dtype=torch.int32)
    plan = torch.zeros((12,), dtype=torch.int64)
    wb = L.g2pc_sampler_bin_table_workspace(HL)
    ws = emu.workspace(wb, "cpu")
    emu.check(L.g2pc_sampler_bin_table(emu.ptr(hist_dev), HL, emu.ptr(stats), int(exact), 1, ops.WAVE_MODE_MIN_DRAWS, emu.ptr(lut),
                                       emu.ptr(quota), emu.ptr(bin_start), emu.ptr(bin_lo), ops.C_void(plan), emu.ptr(ws), wb,
                                       None), "bin_table")
    B, gv, p_wave, any_s, means_rows, rows_ub, err = [int(v) for v in plan[:7]]
    assert err == 0
    hist = np.bincount(ppg, minlength=int(ppg.max()) + 1).astype(np.int64)
    ref = ops.bin_table_from_hist(hist, exact)
    assert B == len(ref)
    got = list(ops._LazyBins(bin_lo, quota, B))
    assert got == [(float(s), float(e), int(n)) for s, e, n in ref]
    # look-up table, members, offsets: the host path's construction
    rl = np.full((int(ppg.max()) + 1,), -1, dtype=np.int32)
    members = np.zeros((B,), dtype=np.int64)
    for b, (s, e, n) in enumerate(ref):
        lo, hi = int(np.ceil(s)), min(int(np.ceil(e)), int(ppg.max()) + 1)
        if n > 0 and hi > lo:
            rl[lo:hi] = b
            members[b] = hist[lo:hi].sum()
    assert np.array_equal(lut.numpy()[:rl.shape[0]], rl) and (lut.numpy()[rl.shape[0]:] == -1).all()
    bs = np.concatenate([[0], np.cumsum(members)])
    assert np.array_equal(bin_start.numpy()[:B + 1].astype(np.int64), bs) and gv == int(bs[-1])
    q = np.array([n for _, _, n in ref])
    wave = [b for b in range(B) if q[b] - 1 >= ops.WAVE_MODE_MIN_DRAWS and members[b] > 0]
    assert p_wave == (int(bs[wave[0]]) if wave else gv)
    assert bool(any_s) == bool(np.any((q > 1) & (members > 0)))
    assert means_rows == int(members[q > 0].sum())
    assert rows_ub == means_rows + int((members * np.maximum(q - 1, 0)).sum())
    lane = [int(q[b]) - 1 for b in range(B) if members[b] > 0 and 0 < q[b] - 1 < ops.WAVE_MODE_MIN_DRAWS]
    assert int(plan[10]) == (max(lane) if lane else 0)            # rows of the lane-mode staging planes (draw-once sampling)


def test_point_counts_beyond_the_device_histogram_fall_back_to_the_host_table(emu, monkeypatch):
    """A Gaussian with more points than the device-side bin table's histogram holds (HIST_GUESS, shrunk here): the plan kernel
    reports it and sample_pointcloud builds the table on the host as before -- same cloud as with the count known up front."""
    import numpy as np
    import torch
    from g2pc import ops
    from g2pc.synth import make_scene
    monkeypatch.setattr(ops, "HIST_GUESS", 64)
    sc = make_scene(300, 21)
    cov, _, nrm = ops.build_covariances(sc.scales, sc.rots, 1.0, want_normals=True)
    ppg = torch.full((300,), 5, dtype=torch.int32)
    ppg[:40] = torch.arange(60, 100, dtype=torch.int32)         # 60 .. 99 points: beyond the 64-entry histogram
    stats = torch.tensor([int(ppg.sum()), 0, 0, int(ppg.max())], dtype=torch.int64)
    kw = dict(exact=False, std=2.0, attempts=5, seed=7)
    a = ops.sample_pointcloud(sc.xyz, cov, sc.colours * 255, nrm, ppg, None, stats=stats, **kw)
    b = ops.sample_pointcloud(sc.xyz, cov, sc.colours * 255, nrm, ppg, int(ppg.max()), **kw)
    assert a.points.shape == b.points.shape and torch.equal(a.points, b.points) and torch.equal(a.colours, b.colours)
    assert list(a.bins) == list(b.bins)
