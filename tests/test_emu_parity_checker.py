"""The benchmark-scale parity checker (tools/parity_cfg2.py) driven through the CPU emulator on the SAME job at a size
the emulator can follow (4 000 Gaussians, cameras 0 and 17 of the 50-camera rig at 320x180, 40 000 points; fixture
tests/golden/*_cfg2_mini.npz written by the untouched reference, oracle/make_golden.py render_mini)."""
import os
import sys

from emu_util import emu  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_parity_checker_on_mini_job(emu):
    import parity_cfg2
    assert parity_cfg2.available("mini")
    r = parity_cfg2.run("cpu", tag="mini")
    print(r)
    assert r["mask_flips"] == 0 and r["culled_equal"] and r["keep_equal"]
    # projected means, radii, depths: every bit of every Gaussian (the fixture's fingerprints), given the reference's cameras
    assert all(k["k1_mismatch"] == 0 and k["radius_mismatch"] == 0 and k["in_mask_flips"] == 0 for k in r["k1"]), r["k1"]
    assert r["contrib_max"] < 1e-5 and r["colour_max"] < 1e-5 and r["image_max"] < 1e-5
    assert r["ppg_mismatch_given_ref_contrib"] == 0
    assert r["sample_points"] == r["sample_points_ref"] and r["sample_rows_unmatched"] == 0
    assert r["sample_xyz_max"] < 1e-4 and r["sample_rgb_max"] < 1e-4
