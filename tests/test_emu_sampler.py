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
