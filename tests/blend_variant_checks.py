"""EXPERIMENTS build only (csrc/experiments/): the blend kernel variants of the python-semantics renderer (g2pc_set_blend_variant) must agree: 1 = dual-list single-wave
kernel (default), 2 / 3 = two-wave form (one wave per 8x8 sub-block, 128-entry batches; unroll 4 / 2), 4 / 5 = scalar-gather form
(A, B, C and colour of a list entry through scalar loads of its record, unroll 4 / 2).  Same loads, tests and
floating-point operations per (pixel, Gaussian) visit, so every contribution at or above the transmittance floor, every
colour and every pixel must be BIT-identical; below the floor the variants may stop a saturated sub-block a batch apart."""
import numpy as np
import torch


def run_variants(device, golden_dir, variants=(1, 2, 3, 4, 5, 6), t_floor=1e-6, pipelined=False):
    from g2pc import _native as nv
    import gauss_render
    from render_checks import run_render_case, assert_render_matches
    out = {}
    try:
        for v in variants:
            nv.experiments().g2pc_set_blend_variant(v)
            gauss_render.clear_context_pool()
            g, R, images, contribs = run_render_case(golden_dir, device=device, t_floor=t_floor)
            assert_render_matches(g, R, images, contribs, max_colour_flips=1)      # (expanded exponent: one swapped arg-max tie in 6 000)
            out[v] = (images, contribs, R.get_gaussian_colours().cpu().numpy(), R.best_key.cpu().numpy().copy())
    finally:
        nv.experiments().g2pc_set_blend_variant(1)
        gauss_render.clear_context_pool()
    return out


def assert_variants_agree(out, t_floor=1e-6):
    base = out[1]
    for v, got in out.items():
        if v == 1:
            continue
        assert np.array_equal(base[0], got[0]), "variant %d: images differ (max %g)" % (v, np.abs(base[0] - got[0]).max())
        above = base[1][-1] >= t_floor
        assert np.array_equal(base[1][:, above], got[1][:, above]), "variant %d: contributions above the floor differ" % v
        assert np.array_equal(base[3][above], got[3][above]), "variant %d: packed keys above the floor differ" % v
        assert np.array_equal(base[2][above], got[2][above]), "variant %d: colours differ" % v
        assert float(np.abs(base[1] - got[1]).max()) <= t_floor
