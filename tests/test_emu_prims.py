"""Kernel LOGIC of the scan / radix sort / geometry / allocation kernels, run through the fiber emulator on
CPU (no GPU here).  The same assertions run on the real MI355X in tests/test_gpu_*.py."""
import os

import numpy as np
import pytest
import torch

from emu_util import emu  # noqa: F401
import ref_gauss as RG
from g2pc import ops
from g2pc.synth import make_scene


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 5000, 1024 * 1024 + 7])
def test_scan(emu, n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 5, size=n).astype(np.int32)
    out = ops.exclusive_scan_u32(torch.from_numpy(v)).numpy()
    ref = np.concatenate([[0], np.cumsum(v)]).astype(np.int64)
    assert np.array_equal(out.astype(np.int64), ref)


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 32), (777, 0, 8), (5000, 0, 13), (20000, 0, 32), (6000, 4, 11), (3000, 0, 3)])
def test_radix_sort_stable(emu, n, lo, hi):
    rng = np.random.default_rng(n + hi)
    keys = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    if hi <= 13:
        keys = (keys % (1 << hi)).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = ops.sort_pairs_u32(torch.from_numpy(keys.view(np.int32)), torch.from_numpy(vals.view(np.int32)), lo, hi)
    ko, vo = ko.numpy().view(np.uint32), vo.numpy().view(np.uint32)
    digit = (keys >> lo) & ((1 << (hi - lo)) - 1)
    order = np.argsort(digit, kind="stable")
    assert np.array_equal(vo, vals[order])
    assert np.array_equal(ko, keys[order])


def test_geometry_matches_oracle(emu, golden_dir):
    g = np.load(os.path.join(golden_dir, "geom_n4096.npz"))
    sc = make_scene(int(g["n"]), int(g["seed"]))
    cov, cov6, nrm = ops.build_covariances(sc.scales, sc.rots, 1.0, want_cov6=True, want_normals=True)
    np.testing.assert_allclose(cov.numpy(), g["cov"], rtol=2e-6, atol=5e-11)   # entries ~1e-4: 5e-7 of scale
    np.testing.assert_array_equal(cov6.numpy(), RG.strip_symmetric(cov).numpy())
    np.testing.assert_allclose(nrm.numpy(), g["normals"], rtol=0, atol=1e-6)
    mags = ops.gaussian_magnitudes(cov, sc.opacities)
    np.testing.assert_allclose(mags.numpy(), g["mags_opacity"], rtol=5e-6)
    bad = torch.from_numpy(g["cov"]).clone()
    bad[torch.from_numpy(g["bad_rows"])] = torch.from_numpy(g["bad_cov"])
    keep = ops.validate_covariances_(bad)
    assert np.array_equal(keep.numpy(), g["keep"])
    np.testing.assert_allclose(bad[keep].numpy(), g["cov_valid"], rtol=1e-6, atol=2e-10)


def test_distribute_points_matches_reference(emu, golden_dir):
    g = np.load(os.path.join(golden_dir, "geom_n4096.npz"))
    mags = torch.from_numpy(g["mags_opacity"])
    ppg64, ppg32, stats = ops.distribute_points(mags, 100000)
    assert np.array_equal(ppg64.numpy(), g["ppg_100k"])
    assert np.array_equal(ppg32.numpy(), g["ppg_100k"].astype(np.int32))
    assert int(stats[3]) == int(g["ppg_100k"].max())
    over, over32, st = ops.distribute_points(torch.tensor([1.5] * 4 + [0.01] * 6, dtype=torch.float64), 7)
    assert np.array_equal(over.numpy(), g["ppg_overshoot"])            # negative-slice quirk
    assert int(st[2]) == -1 and int(st[1]) == 6
    hist = ops.bincount(ppg32, int(stats[3]) + 1).numpy()
    assert np.array_equal(hist, np.bincount(g["ppg_100k"].astype(np.int64)))
    assert ops.calculate_bin_sizes_from_hist(hist) == (int(g["start_bin"]), int(g["bin_size"]))
    # underfill: every zero entry gets a point while the budget lasts, in index order
    sizes = torch.tensor([10.0, 0.001, 0.001, 10.0, 0.001], dtype=torch.float64)
    p, _, st = ops.distribute_points(sizes, 22)
    assert p.tolist() == RG.distribute_points(sizes, 22).tolist()


def test_dpp_wave_reductions(emu):
    import ctypes
    from g2pc import _native as nv
    rng = np.random.default_rng(3)
    v = rng.integers(0, 2 ** 32, size=64 * 37, dtype=np.uint64).astype(np.uint32)
    v[64:128] = 7
    t = torch.from_numpy(v.view(np.int32))
    out = torch.zeros(4 * 37, dtype=torch.int32)
    nv.check(nv.lib().g2pc_selftest_wave_reduce(nv.ptr(t), nv.ptr(out), 37, None), "selftest")
    o = out.numpy().view(np.uint32).reshape(37, 4)
    r = v.reshape(37, 64)
    assert np.array_equal(o[:, 0], r.max(1)) and np.array_equal(o[:, 1], r.min(1))
    assert np.array_equal(o[:, 2], r.max(1)) and np.array_equal(o[:, 3], r.min(1))


def _bucket_sort(nv, keys, vals):
    import ctypes as C
    L = nv.lib()
    n = keys.numel()
    ko, vo = torch.empty_like(keys), torch.empty_like(keys)
    flag = torch.zeros(1, dtype=torch.int32, device=keys.device)
    wb = L.g2pc_bucket_sort_workspace(n)
    ws = torch.empty(wb, dtype=torch.uint8, device=keys.device)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    nv.check(L.g2pc_bucket_sort_u32(p(keys), p(vals), p(ko), p(vo), n, p(flag), p(ws), wb, nv.stream_handle(keys.device)), "bucket_sort")
    return ko, vo, int(flag.item())


@pytest.mark.parametrize("n,ties", [(1, 1), (63, 6), (1000, 100), (40_000, 500), (40_001, 950), (700_000, 500)])
def test_bucket_sort_equals_stable_argsort(emu, n, ties):
    """Depth-like keys (bit patterns of positive floats in a narrow range, with exact ties and 'off screen' sentinels):
    the bucket sort returns the stable ascending order, sentinels last."""
    rng = np.random.default_rng(n)
    depth = rng.uniform(2.5, 4.5, size=n).astype(np.float32)
    depth[rng.integers(0, n, size=ties)] = depth[0]       # exact ties in ONE bucket (fewer than its room: 65 .. 1024 items walk the R = 1 .. 16 register sorts)
    keys = depth.view(np.uint32).copy()
    off = rng.random(n) < 0.1
    keys[off] = 0xFFFFFFFF
    vals = rng.permutation(n).astype(np.int32)
    ko, vo, flag = _bucket_sort(emu, torch.from_numpy(keys.view(np.int32)), torch.from_numpy(vals))
    assert flag == 0
    order = np.argsort(keys, kind="stable")
    nvalid = int((~off).sum())
    assert np.array_equal(ko.numpy().view(np.uint32), keys[order])
    assert np.array_equal(vo.numpy()[:nvalid], vals[order][:nvalid])              # sorted part: exact stable order
    assert sorted(vo.numpy()[nvalid:].tolist()) == sorted(vals[order][nvalid:].tolist())    # tail: any order
    # values omitted: the sort returns the input positions themselves (no index array is read)
    ko2, vo2, flag2 = _bucket_sort(emu, torch.from_numpy(keys.view(np.int32)), None)
    assert flag2 == 0 and np.array_equal(ko2.numpy().view(np.uint32), keys[order])
    assert np.array_equal(vo2.numpy()[:nvalid], order[:nvalid].astype(np.int32))


def test_bucket_sort_flags_a_pile_up(emu):
    """More than 4096 keys in 1/1024 of the key range: the overflow flag is raised (the caller falls back to the radix sort)."""
    n = 20_000
    keys = np.full(n, np.float32(3.0)).view(np.uint32).copy()
    keys[0] = np.float32(2.5).view(np.uint32)
    keys[1] = np.float32(4.5).view(np.uint32)
    ko, vo, flag = _bucket_sort(emu, torch.from_numpy(keys.view(np.int32)), torch.arange(n, dtype=torch.int32))
    assert flag > 4096
    assert sorted(vo.numpy().tolist()) == list(range(n))                          # still a permutation
