"""EXPERIMENTS build (csrc/experiments/blend_variants.inl; not part of libg2pc.so): the two-wave and scalar-gather forms of the
dual-list blend against the single-wave kernel and the reference's golden vectors, on the CPU emulator (real barriers between
the two waves of a block); the scalar blend kernel for 1 and 4 sub-blocks per wave against the reference's golden vectors."""
import pytest

from emu_util import emu, emu_exp  # noqa: F401


def test_two_wave_blend_equals_single_wave_blend(emu_exp, golden_dir):
    from blend_variant_checks import run_variants, assert_variants_agree
    # (three of the six kernels of the experiments build -- single wave, two waves, dual list --: the product blends with the dual-list
    # kernel only; all six against each other: tools/experiments/blend_variant_fuzz.py)
    assert_variants_agree(run_variants("cpu", golden_dir, variants=(1, 3, 6)))


@pytest.mark.parametrize("sub", [4, 1])
def test_scalar_blend_matches_reference_python_renderer(emu_exp, golden_dir, monkeypatch, sub):
    """k_blend_py<4, 1> / <1, 4> (4 / 1 sub-blocks per wave) against the outputs of the untouched reference."""
    import gauss_render
    from render_checks import run_render_case, assert_render_matches
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", sub)
    g, R, images, contribs = run_render_case(golden_dir)
    print(sub, assert_render_matches(g, R, images, contribs, max_colour_flips=0))


def test_product_library_refuses_other_sub_block_counts(emu, monkeypatch):
    """libg2pc.so blends layouts of 2 sub-blocks per wave only (ABI 6) and has no knobs: it says so instead of guessing."""
    import gauss_render
    import camera_handler
    import torch
    from g2pc import _native as nv
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    assert not nv.has_experiments()
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", 4)
    sc = make_scene(50, 3, scale_lo=0.02, scale_hi=0.05)
    tr, intr = make_cameras(1, width=64, height=40, focal=56.0)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances)
    k = next(iter(tr))
    with pytest.raises(nv.G2pcError, match="chunk_subblocks must be 2"):
        R(camera_handler.get_camera("python", torch.tensor(tr[k]), intr[k]))
    gauss_render.clear_context_pool()
