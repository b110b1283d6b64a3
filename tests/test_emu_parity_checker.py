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


def test_row_matching_survives_shifted_order():
    """An accept/reject flip removes one row of ours and adds another elsewhere: every later row sits one place off.  The
    matched-row comparison must still pair every reference row with its point and compare THAT row's colour."""
    import numpy as np
    import parity_cfg2
    rng = np.random.default_rng(5)
    ref = rng.random((6400, 3)).astype(np.float32)
    rgb = (rng.random((6400, 3)) * 255).astype(np.float32)
    ours = np.delete(ref, 1000, axis=0)                     # a point lost ...
    ours = np.insert(ours, 5000, [[9, 9, 9]], axis=0)       # ... and one gained further on: rows 1000..4999 are shifted by one
    ours_rgb = np.insert(np.delete(rgb, 1000, axis=0), 5000, [[1, 2, 3]], axis=0)
    rr = np.arange(0, 6400, 64)
    r = parity_cfg2.match_rows(ours, ours_rgb, ref[::64], rgb[::64], rr)
    assert r["sample_rows_unmatched"] == 0 and r["sample_xyz_max"] == 0.0 and r["sample_rgb_max"] == 0.0, r
    assert r["sample_rows_order_shifted"]["count"] == len([x for x in rr if 1000 < x <= 5000])
    assert r["sample_rows_order_shifted"]["first_row"] == 1024 and r["sample_rows_order_shifted"]["max_offset"] == 1
    bad = ours_rgb.copy()
    bad[2047] += 3.0                                        # the matched row of reference row 2048
    assert parity_cfg2.match_rows(ours, bad, ref[::64], rgb[::64], rr)["sample_rgb_max"] > 1e-2
    # a reference row with no counterpart at all is reported, not silently dropped
    gone = np.delete(ours, 2047, axis=0)
    assert parity_cfg2.match_rows(gone, np.delete(ours_rgb, 2047, axis=0), ref[::64], rgb[::64], rr)["sample_rows_unmatched"] == 1


def test_all_camera_production_checker_on_mini_job(emu, monkeypatch):
    """tools/parity_all50.py (the 50-camera production-path checker behind bench.py's `parity` block) on a rehearsal of the same
    generator at a size the emulator follows: 4 000 Gaussians, all 6 cameras of a 6-camera rig at 320x180, 40 000 points
    (tests/golden/render_py_cfg2_mini_all6.npz, oracle/make_golden.py render_all_mini), pipelined cameras included."""
    import gauss_render
    import parity_all50
    gauss_render.clear_context_pool()
    monkeypatch.setattr(gauss_render, "PIPELINE_IN_EMULATOR", True)
    monkeypatch.setattr(gauss_render, "PIPELINE_STREAMS", 2)
    assert parity_all50.available("mini_all6")
    r = parity_all50.run("cpu", tag="mini_all6")
    print(r)
    gauss_render.clear_context_pool()
    assert r["mask_flips"] == 0 and r["culled_equal"] and r["keep_equal"]
    assert r["contrib_max"] < 1e-5 and r["winner_camera_mismatch"] == 0
    # a Gaussian's colour is the colour of its arg-max PIXEL: where two pixels tie to ~1e-6 the floor mode's expanded exponent may
    # pick the other one (contributions unaffected) -- counted, and absent in the to-the-letter mode below
    assert r["colour_off_gaussians"] <= 2
    assert r["ppg_mismatch_given_ref_contrib"] == 0 and r["ppg_mismatch_end_to_end"] == r["ppg_flips_explained"]
    assert r["sample_points"] == r["sample_points_ref"] and r["sample_rows_unmatched"] == 0
    assert r["sample_xyz_max"] < 1e-4 and r["sample_rgb_max"] < 1e-4
    e = parity_all50.run("cpu", tag="mini_all6", sampler=False, t_floor=0.0)
    gauss_render.clear_context_pool()
    assert e["mask_flips"] == 0 and e["contrib_max"] < 1e-5 and e["winner_camera_mismatch"] == 0
    assert e["colour_off_gaussians"] == 0 and e["colour_max"] < 1e-5, e
