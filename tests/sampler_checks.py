"""Shared sampler parity assertions (used by the emulator tests on CPU and the -m gpu tests)."""
import os

import numpy as np
import torch

import ref_gauss as RG
from g2pc import ops
from g2pc.synth import make_scene


def run_sampler_case(golden_dir, name, device="cpu"):
    g = np.load(os.path.join(golden_dir, name))
    sc = make_scene(int(g["n"]), int(g["seed"]))
    dev = torch.device(device)
    cov, _, nrm = ops.build_covariances(sc.scales.to(dev), sc.rots.to(dev), 1.0, want_normals=True)
    keep = ops.validate_covariances_(cov)
    assert bool(keep.all())
    np.testing.assert_allclose(cov.cpu().numpy(), g["cov_valid"], rtol=2e-6, atol=5e-10)
    # the allocation is checked bit-exactly on the reference's own covariances (isolates sampling
    # from 1-ulp differences of the covariance build)
    cov_ref = torch.from_numpy(g["cov_valid"]).to(dev)
    mags = ops.gaussian_magnitudes(cov_ref, sc.opacities.to(dev))
    ppg64, ppg32, stats = ops.distribute_points(mags, int(g["num_points"]))
    exact = bool(g["exact"])
    out = ops.sample_pointcloud(sc.xyz.to(dev), cov_ref, (sc.colours * 255).to(dev), nrm, ppg32, int(stats[3]),
                                exact=exact, std=2.0, attempts=100 if exact else 5, seed=int(g["noise_seed"]),
                                want_index=True)
    return g, sc, ppg32.cpu().numpy(), out


def assert_sampler_matches(g, ppg, out, max_ppg_mismatch=0):
    # 1) points-per-Gaussian: bit exact
    bad = int((ppg != g["ppg"]).sum())
    assert bad <= max_ppg_mismatch, "ppg mismatches: %d" % bad
    pts, ref = out.points.cpu().numpy(), g["points"]
    # 2) same number of points, same order, xyz within 1e-4 (north_star tolerance); report the exact count
    assert pts.shape == ref.shape, (pts.shape, ref.shape)
    d = np.abs(pts - ref).max(axis=1)
    n_off = int((d > 1e-4).sum())
    assert n_off == 0, "points off by more than 1e-4: %d (max %g)" % (n_off, d.max())
    assert d.max() < 2e-6                      # in practice: fp32 rounding of the Cholesky / Box-Muller only
    np.testing.assert_allclose(out.colours.cpu().numpy(), g["colours"], atol=1e-4)
    if out.normals is not None and "normals" in g:
        np.testing.assert_allclose(out.normals.cpu().numpy(), g["normals"], atol=1e-6)
