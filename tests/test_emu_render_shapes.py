"""Odd shapes through the python-semantics renderer (emulator) against the oracle: image sizes that are not multiples of
the 8x8 sub-blocks, a single Gaussian, nothing in front of the camera, splats larger than the image, every pixels-per-lane
variant, with and without the transmittance floor."""
import numpy as np
import pytest
import torch

from emu_util import emu, emu_exp  # noqa: F401


@pytest.mark.parametrize("n,w,h,scale,sub,floor", [
    (1, 64, 48, (0.05, 0.06), 2, 0.0),            # one Gaussian
    (7, 17, 9, (0.02, 0.2), 2, 0.0),              # tiny image, sub-blocks mostly padding
    (120, 61, 61, (0.3, 0.6), 2, 1e-6),           # one pixel over the 60-pixel tile limit; splats larger than the image
    (300, 123, 77, (0.005, 0.05), 2, 1e-6),       # odd sizes
    (300, 200, 50, (0.005, 0.05), 2, 0.0),        # wide image, packed kernel, exact mode
])
def test_shapes_vs_oracle(emu, monkeypatch, n, w, h, scale, sub, floor):
    import gauss_render
    from render_checks import run_vs_oracle
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", sub)
    res = run_vs_oracle(n, 100 + n, w, h, 0.9 * w, 2, scale=scale, t_floor=floor)
    assert res["image"] < 1e-4 and res["contribution"] < 1e-4 and res["colour"] < 1e-4 and res["flips"] == 0, res


def test_more_than_4096_leaf_tiles(emu, monkeypatch):
    """16 384 quad-tree leaves (640 x 400 at max_tile_size 5 -- what 7680 x 4320 is at the default 60): the packed keys widen
    their tile field to 14 bits (63 cameras per key epoch).  Image, contributions, colours and visibility against the oracle
    (reference: gauss_render.py:290-335 with render()'s max_tile_size).  This sparse scene (400 Gaussians under 16 384
    leaves) also meets the reference's empty-node rule: an interior quad-tree node that holds no Gaussian is painted with the
    background and its children are never visited (gauss_render.py:311-314), while a child reaches one pixel beyond an
    odd-sized parent (ceil / floor at :321-334) -- a Gaussian that reaches only into that pixel is skipped with the leaf
    (k_tile_gate / _static_plan; the leaf grid alone left 5e-4 of the image's pixels off by up to 2e-3 here)."""
    import gauss_render
    from render_checks import run_vs_oracle
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", 2)
    res = run_vs_oracle(400, 515, 640, 400, 600.0, 2, scale=(0.004, 0.05), t_floor=1e-6, max_tile_size=5)
    assert res["seq_bits"] == 14
    assert res["contribution"] < 1e-5 and res["flips"] == 0, res
    assert res["image"] < 1e-5 and res["image_frac_off"] == 0.0, res
    assert res["colour_off_gaussians"] <= 3, res       # (arg-max pixels tying to ~1e-6 under the default floor, see render_checks)


def test_nothing_in_front_of_the_camera(emu):
    """Every Gaussian behind the camera: zero instances, background image, no contribution, no visible Gaussian."""
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    sc = make_scene(50, 3, scale_lo=0.01, scale_hi=0.05)
    tr, intr = make_cameras(1, width=64, height=40, focal=60.0)
    name = next(iter(tr))
    c2w = torch.tensor(tr[name])
    xyz = sc.xyz * 0.1 + c2w[:3, 3] + 2.0 * c2w[:3, 2]          # the camera looks down -z: +z is behind it
    G = Gaussians(xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
    img = R(camera_handler.get_camera("python", c2w, intr[name]))[0]
    assert R.last_stats[-1][0] == 0
    assert torch.equal(img, torch.ones_like(img))                # white background
    assert float(R.gaussian_max_contribution.max()) == 0.0 and int(R.get_visible_gaussians().sum()) == 0



@pytest.mark.parametrize("sub", [1, 2])
def test_floor_mode_keeps_contributions_bit_identical(emu, request, monkeypatch, sub):
    """The documented guarantee of the transmittance floor: the floor only stops walks whose remaining contributions are all
    below it.  With one 8x8 sub-block per wave both modes run the same kernel and every contribution above the floor equals
    the exact mode's bit for bit, as do the winners.  With two (the default) the exact mode runs k_blend_py_pk -- the
    reference's operation order in the exponent -- and the floor mode k_blend_py_dl -- the exponent expanded about the
    sub-block centre, <= 2e-5 relative in alpha: contributions agree to a few 1e-6, an arg-max may move between pixels whose
    contributions tie at that level."""
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    from g2pc.synth import make_scene, make_cameras
    if sub == 1:
        request.getfixturevalue("emu_exp")         # (one sub-block per wave: the scalar blend of the experiments build)
    monkeypatch.setattr(gauss_render, "BLEND_SUBBLOCKS", sub)
    sc = make_scene(1500, 41, scale_lo=0.004, scale_hi=0.03)
    tr, intr = make_cameras(2, width=160, height=90, focal=140.0)
    G = Gaussians(sc.xyz, sc.scales, sc.rots, sc.colours, sc.opacities)
    out = []
    for floor in (0.0, 1e-6):
        R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances,
                                      visible_gaussian_threshold=0.05)
        R.t_floor = floor
        imgs = [R(camera_handler.get_camera("python", torch.tensor(tr[k]), intr[k]))[0].numpy() for k in tr]
        out.append((R.gaussian_max_contribution.numpy().copy(), R.get_gaussian_colours().numpy().copy(), np.stack(imgs),
                    R.best_key.numpy().copy()))
    exact, floored = out
    seen = exact[0] >= 1e-5
    assert seen.sum() > 300
    if sub == 1:
        assert np.array_equal(exact[0][seen], floored[0][seen])                    # contributions, bit for bit
        assert np.array_equal(exact[3][seen], floored[3][seen])                    # and the winning (camera, tile, pixel)
        assert np.abs(exact[1][seen] - floored[1][seen]).max() < 1e-3              # winners' colours (0..255 scale)
        assert np.abs(exact[2] - floored[2]).max() < 2e-6                          # images
    else:
        assert np.abs(exact[0][seen] - floored[0][seen]).max() < 5e-6
        same_winner = ((exact[3] & 0xFFFFFFFF) == (floored[3] & 0xFFFFFFFF))[seen]   # low word = ~(camera, tile, pixel)
        assert same_winner.mean() > 0.995
        assert np.abs(exact[1][seen] - floored[1][seen])[same_winner].max() < 2e-3
        assert np.abs(exact[2] - floored[2]).max() < 1e-5
