"""Shared renderer parity assertions against the reference python renderer's golden outputs
(tests/golden/render_py_n6000.npz, written by oracle/make_golden.py from the untouched reference)."""
import os

import numpy as np
import torch

from g2pc.synth import make_scene, make_cameras


def run_render_case(golden_dir, device="cpu", t_floor=0.0):
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    g = np.load(os.path.join(golden_dir, "render_py_n6000.npz"))
    dev = torch.device(device)
    sc = make_scene(int(g["n"]), int(g["seed"]), scale_lo=float(g["scale_lo"]), scale_hi=float(g["scale_hi"]))
    transforms, intr = make_cameras(int(g["ncam"]), width=int(g["width"]), height=int(g["height"]), focal=float(g["focal"]))
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    R = gauss_render.get_renderer("python", G.xyz, torch.unsqueeze(torch.clone(G.opacities), 1), G.colours,
                                  G.covariances, visible_gaussian_threshold=0.05)
    R.t_floor = t_floor
    images, contribs = [], []
    for name in transforms:
        cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=None)
        img, _, _, _ = R(cam)
        images.append(img.cpu().numpy())
        contribs.append(R.gaussian_max_contribution.cpu().numpy())
    return g, R, np.stack(images), np.stack(contribs)


def assert_render_matches(g, R, images, contribs, tol=1e-4, allow_mask_flips=0, max_colour_flips=0):
    # images within tol
    d_img = np.abs(images - g["images"]).max()
    assert d_img < tol, "image max abs diff %g" % d_img
    # running max contribution after every camera
    d_c = np.abs(contribs - g["contrib_after_cam"]).max()
    assert d_c < tol, "contribution max abs diff %g" % d_c
    # per-Gaussian colours (0..255 scale in the API -> compare in 0..1)
    cols = R.get_gaussian_colours().cpu().numpy() / 255.0
    ref_cols = g["colours"] / 255.0
    # a Gaussian's colour IS the rendered colour of its arg-max pixel: two pixels whose contributions tie to ~1e-6 may swap
    # under any change of rounding (image and contributions, checked above, are unaffected) -> count such Gaussians
    off = np.abs(cols - ref_cols) >= tol
    n_off = int(off.any(axis=1).sum())
    d_col = float(np.abs(cols - ref_cols)[~off].max()) if (~off).any() else 0.0
    assert n_off <= max_colour_flips, "%d Gaussians with another arg-max pixel's colour (max abs diff %g)" % (
        n_off, np.abs(cols - ref_cols).max())
    # culling mask: exact, report the margin of anything that flips
    vis = R.get_visible_gaussians().cpu().numpy()
    flips = np.nonzero(vis != g["visible"])[0]
    margins = np.abs(g["contrib_after_cam"][-1][flips] - 0.05)
    assert len(flips) <= allow_mask_flips, "visible-mask flips: %d, margins %s" % (len(flips), margins)
    np.testing.assert_allclose(R.get_total_gaussian_contributions().cpu().numpy(), g["total"], atol=tol)
    return dict(image=d_img, contribution=d_c, colour=d_col, colour_flips=n_off, flips=len(flips))


def run_vs_oracle(n, seed, width, height, focal, ncam, device="cpu", scale=(0.004, 0.04), colour_resolution=None,
                  t_floor=0.0, max_tile_size=None, max_gaussians_per_tile=None, xyz_scale=1.0, pipelined=False):
    """HIP renderer vs oracle/ref_render.py (itself bit-pinned to the reference) on a seeded synthetic scene."""
    import gauss_render
    import camera_handler
    import ref_gauss as RG
    import ref_render as RR
    from gauss_handler import Gaussians
    dev = torch.device(device)
    sc = make_scene(n, seed, scale_lo=scale[0], scale_hi=scale[1])
    if xyz_scale != 1.0:                           # crowd the scene towards its centre (leaves over max_gaussians_per_tile)
        sc = sc._replace(xyz=sc.xyz * xyz_scale)
    transforms, intr = make_cameras(ncam, width=width, height=height, focal=focal)
    G = Gaussians(sc.xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    # both renderers get the SAME host-arithmetic inputs: the oracle's covariances (torch.exp on the CPU is MKL's, within an
    # ulp of the library's correctly rounded one) and, below, its camera matrices
    cov = RG.covariances(sc.scales, sc.rots)
    assert float((G.covariances.cpu() - cov).abs().max()) <= 1e-6 * float(cov.abs().max())
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, cov.to(dev),
                                  visible_gaussian_threshold=0.05)
    R.t_floor = t_floor
    if max_tile_size is not None:                  # render()'s max_tile_size argument (gauss_render.py:266), both sides
        R.MAX_TILE_SIZE = max_tile_size
    okw = {} if max_tile_size is None else {"max_tile_size": max_tile_size}
    if max_gaussians_per_tile is not None:         # ... and its max_gaussians_per_tile (the count-driven split, :319)
        R.MAX_GAUSSIANS_PER_TILE = max_gaussians_per_tile
        okw["max_gaussians_per_tile"] = max_gaussians_per_tile
    O = RR.PythonRendererOracle(sc.xyz, sc.opacities.unsqueeze(1), sc.colours.double(), cov, threshold=0.05, **okw)
    worst = dict(image=0.0, contribution=0.0, colour=0.0, flips=0, near_threshold=0, image_frac_off=0.0)
    for name in transforms:
        cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=colour_resolution)
        ocam = RR.get_camera(torch.tensor(transforms[name]), intr[name], colour_resolution=colour_resolution)
        assert torch.allclose(cam.world_view_transform, ocam.world_view_transform, rtol=1e-5, atol=1e-6)
        cam.world_view_transform, cam.projection_matrix = ocam.world_view_transform, ocam.projection_matrix
        cam.FoVx, cam.FoVy, cam.focal_x, cam.focal_y = ocam.FoVx, ocam.FoVy, ocam.focal_x, ocam.focal_y
        ref = O(ocam)
        if pipelined:                              # the capture / replay path returns no image
            R(cam, return_image=False)
            continue
        img = R(cam)[0].cpu()
        d = (img - ref).abs()
        worst["image"] = max(worst["image"], float(d.max()))
        worst["image_frac_off"] = max(worst["image_frac_off"], float((d > 1e-4).float().mean()))
    c = R.gaussian_max_contribution.cpu()
    dc = (c - O.max_contribution).abs()
    worst["contribution"] = float(dc.max())
    worst["contribution_frac_off"] = float((dc > 1e-4).float().mean())
    dcol = (R.get_gaussian_colours().cpu().double() - O.get_gaussian_colours()).abs() / 255.0
    # colours are compared where the reference's contribution is a normal float: a DENORMAL contribution (< 1.2e-38, a product
    # T * alpha of a crowded leaf that underflowed) still counts as "seen" in the reference's strict `>` against 0, while the
    # blend's candidate filter starts at FLT_MIN -- such a Gaussian carries a colour there and none here (colour_off_tiny)
    seen = O.max_contribution > max(t_floor, 1e-30)
    tiny = (O.max_contribution <= 1e-30) & (O.max_contribution >= (t_floor if t_floor > 0 else 0.0))
    worst["colour_off_tiny"] = int((dcol[tiny] > 1e-4).any(dim=1).sum())
    worst["colour"] = float(dcol[seen].max()) if bool(seen.any()) else 0.0
    worst["colour_frac_off"] = float((dcol[seen] > 1e-4).float().mean()) if bool(seen.any()) else 0.0
    # Gaussians whose colour is off: each is one flipped arg-max between pixels whose contributions tie to ~1e-6 (the
    # colour is the rendered colour of the winning pixel), image and contributions unaffected
    worst["colour_off_gaussians"] = int((dcol[seen] > 1e-4).any(dim=1).sum())
    worst["seen_gaussians"] = int(seen.sum())
    flips = (R.get_visible_gaussians().cpu() != O.get_visible_gaussians())
    worst["flips"] = int(flips.sum())
    worst["seq_bits"] = R.seq_bits
    worst["split_leaves"] = R.split_leaves
    worst["host_driven"], worst["child_pass_cameras"] = R.host_driven, R.child_pass_cameras
    worst["near_threshold"] = int(((O.max_contribution - 0.05).abs() < 1e-5).sum())
    worst["flip_margins"] = (O.max_contribution[flips] - 0.05).abs().tolist()
    return worst


def unpack_keys(keys, seq_bits):
    """(contribution bits, camera slot, tile sequence number, pixel) of packed visibility keys ([n, 4] uint64).  The pipeline
    reserves room for the children of an on-demand child pass in the tile field (5 x leaves: 13 bits at 1280 x 720), the
    two-call path widens the field only when it must -- the same keys in two packings."""
    k = np.ascontiguousarray(keys).view(np.uint64)
    order = (~k & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    seen = (k >> np.uint64(32)) != 0
    return np.stack([(k >> np.uint64(32)), np.where(seen, order >> np.uint64(12 + seq_bits), 0),
                     np.where(seen, (order >> np.uint64(12)) & np.uint64((1 << seq_bits) - 1), 0),
                     np.where(seen, order & np.uint64(0xFFF), 0)], axis=1)


def run_split_fixture(golden_dir, tag, device="cpu", t_floor=0.0, pipelined=False):
    """HIP renderer against tests/golden/render_py_split_<tag>.npz -- outputs of the untouched reference on a scene whose leaves
    exceed max_gaussians_per_tile (oracle/make_golden.py render_split)."""
    import gauss_render
    import camera_handler
    from gauss_handler import Gaussians
    g = np.load(os.path.join(golden_dir, "render_py_split_%s.npz" % tag))
    dev = torch.device(device)
    sc = make_scene(int(g["n"]), int(g["seed"]), scale_lo=float(g["scale_lo"]), scale_hi=float(g["scale_hi"]))
    xyz = sc.xyz * float(g["crowd"])
    transforms, intr = make_cameras(1, width=int(g["width"]), height=int(g["height"]), focal=float(g["focal"]))
    name = next(iter(transforms))
    G = Gaussians(xyz.to(dev), sc.scales.to(dev), sc.rots.to(dev), sc.colours.to(dev), sc.opacities.to(dev))
    R = gauss_render.get_renderer("python", G.xyz, G.opacities.unsqueeze(1), G.colours, G.covariances, visible_gaussian_threshold=0.05)
    R.t_floor = t_floor
    pin = int(g["tile_pin"])
    R.MAX_GAUSSIANS_PER_TILE, R.MAX_TILE_SIZE = pin, pin // 1000          # gauss_render.py:440-444 under the fixture's memory pin
    cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=None)
    res = dict(image=0.0)
    if pipelined:
        R(cam, return_image=False)                                         # (first camera of a job: learns the capacity)
        R(cam, return_image=False)                                         # the same camera again through the graph: same maxima
    else:
        res["image"] = float(np.abs(R(cam)[0].cpu().numpy() - g["image"]).max())
    c = R.gaussian_max_contribution.cpu().numpy()
    res["contribution"] = float(np.abs(c - g["contrib"]).max())
    cols = R.get_gaussian_colours().cpu().numpy().astype(np.float64)
    stride = int(g["stride"])
    # colours are compared where the reference's contribution is a normal float well above underflow: below ~1e-30 the products
    # T * alpha of a crowded leaf are denormal or flushed to zero depending on the host's / device's exp and multiply, and a
    # Gaussian the reference leaves at "never seen" (0, no colour) may carry a 1e-40 contribution and a colour here, or the
    # other way round (reported as colour_off_tiny; no threshold a user could set separates them)
    cs = g["contrib"][::stride]
    seen = cs > max(t_floor, 1e-30)
    dall = np.abs(cols[::stride] - g["colours"]) / 255.0
    dcol = dall[seen]
    res["colour"] = float(dcol.max()) if dcol.size else 0.0
    res["colour_off_gaussians"] = int((dcol > 1e-4).any(axis=1).sum())
    res["colour_off_tiny"] = int((dall[~seen & (cs >= (t_floor if t_floor > 0 else 0.0))] > 1e-4).any(axis=1).sum())
    res["tiny"] = int((~seen).sum())
    vis = np.unpackbits(g["visible"])[:int(g["n"])].astype(bool)
    res["flips"] = int((R.get_visible_gaussians().cpu().numpy() != vis).sum())
    res["near_threshold"] = int((np.abs(g["contrib"] - 0.05) < 1e-5).sum())
    res["split_leaves"] = R.split_leaves
    res["host_driven"], res["child_pass_cameras"] = R.host_driven, R.child_pass_cameras
    return res


def run_hidden_behind_wall(device="cpu", threshold=0.0, t_floor=None):
    """Visibility at the API's default threshold (0.0, gauss_render.py:467-468): `get_visible_gaussians` is a strict `>`
    against the running maximum (:249-252, :387), so a Gaussian ALL of whose contributions lie in (0, 2^-25) -- here: small
    Gaussians behind 66 translucent layers that cover the image, transmittance ~1e-10 -- is visible and coloured in the reference.
    Returns the masks of the renderer under test (constructed through get_renderer with `threshold`, floor left to the
    renderer unless given) and of the oracle, plus the oracle's contributions."""
    import gauss_render
    import camera_handler
    import ref_gauss as RG
    import ref_render as RR
    dev = torch.device(device)
    g = torch.Generator().manual_seed(11)
    transforms, intr = make_cameras(1, width=96, height=64, focal=80.0)       # one camera at (3.5, 0, 0) looking down -x
    nw = 66                               # more layers than one 64-entry batch of a tile's list: a floor ends the walk before them
    wall_x = 1.0 - 0.01 * torch.arange(nw)
    wall = torch.stack([wall_x, torch.zeros(nw), torch.zeros(nw)], dim=1)
    hidden = torch.cat([torch.full((40, 1), -0.5), (torch.rand((40, 2), generator=g) - 0.5) * 0.3], dim=1)
    free = torch.cat([torch.full((24, 1), 1.5), (torch.rand((24, 2), generator=g) - 0.5) * 1.6], dim=1)  # in front of the wall
    xyz = torch.cat([wall, hidden, free]).float()
    n = xyz.shape[0]
    scales = torch.log(torch.cat([torch.full((nw, 3), 6.0), torch.full((40, 3), 0.02), torch.full((24, 3), 0.03)])).float()
    rots = torch.tensor([[1.0, 0.0, 0.0, 0.0]]).repeat(n, 1)
    opac = torch.cat([torch.full((nw,), 0.3), torch.full((40,), 0.8), torch.full((24,), 0.6)]).float()
    colours = torch.rand((n, 3), generator=g).float()
    cov = RG.covariances(scales, rots)
    R = gauss_render.get_renderer("python", xyz.to(dev), opac.unsqueeze(1).to(dev), colours.to(dev), cov.to(dev),
                                  visible_gaussian_threshold=threshold)
    if t_floor is not None:
        R.t_floor = t_floor
    O = RR.PythonRendererOracle(xyz, opac.unsqueeze(1), colours.double(), cov, threshold=threshold)
    name = next(iter(transforms))
    cam = camera_handler.get_camera("python", torch.tensor(transforms[name]), intr[name], colour_resolution=None)
    ocam = RR.get_camera(torch.tensor(transforms[name]), intr[name], colour_resolution=None)
    cam.world_view_transform, cam.projection_matrix = ocam.world_view_transform, ocam.projection_matrix
    cam.FoVx, cam.FoVy, cam.focal_x, cam.focal_y = ocam.FoVx, ocam.FoVy, ocam.focal_x, ocam.focal_y
    R(cam)
    O(ocam)
    oc = O.max_contribution
    dcol = ((R.get_gaussian_colours().cpu().double() - O.get_gaussian_colours()).abs() / 255.0).max(dim=1)[0]
    return dict(t_floor=R.t_floor, mask=R.get_visible_gaussians().cpu(), ref_mask=O.get_visible_gaussians(), ref_contrib=oc,
                contrib=R.gaussian_max_contribution.cpu(), colour_err=dcol,
                hidden_tiny=int(((oc > 0) & (oc < 2.0 ** -25))[nw:nw + 40].sum()))


def assert_hidden_behind_wall(device="cpu"):
    # threshold 0 (the API default): the renderer must pick the to-the-letter blend by itself and agree with the reference
    r = run_hidden_behind_wall(device, threshold=0.0)
    assert r["hidden_tiny"] >= 20, r["hidden_tiny"]              # the scene does hold Gaussians with contributions in (0, 2^-25)
    assert r["t_floor"] == 0.0
    assert bool((r["mask"] == r["ref_mask"]).all()), (r["mask"] != r["ref_mask"]).nonzero().flatten().tolist()
    seen = r["ref_contrib"] > 1e-30
    assert float(r["colour_err"][seen].max()) < 1e-4
    rel = ((r["contrib"] - r["ref_contrib"]).abs() / r["ref_contrib"].clamp(min=1e-30))[seen]
    assert float(rel.max()) < 1e-3, float(rel.max())              # (relative: the contributions at stake are ~1e-12)
    # a threshold below the default floor as well
    r = run_hidden_behind_wall(device, threshold=1e-7)
    assert r["t_floor"] == 0.0 and bool((r["mask"] == r["ref_mask"]).all())
    # above the floor the default floor stays on and the mask is the reference's too
    r = run_hidden_behind_wall(device, threshold=0.05)
    assert r["t_floor"] > 0.0 and bool((r["mask"] == r["ref_mask"]).all())
    # ... and this is the hole the rule closes: the floor FORCED on at threshold 0 loses the hidden Gaussians
    r = run_hidden_behind_wall(device, threshold=0.0, t_floor=1e-6)
    assert int((r["mask"] != r["ref_mask"]).sum()) >= 20
