"""The size-driven quad-tree as tile layouts (g2pc/tiles.py) against the oracle's queue (oracle/ref_render.py::quadtree_leaves,
a restatement of gauss_render.py:290-335 pinned bit for bit to the reference) when every node holds a Gaussian: the leaves of
the layout, plus -- for image sizes whose tree is not of uniform depth -- the children of its `tile_force` nodes level by
level (tiles.child_layout), must be the oracle's leaves in the oracle's FIFO order.  Pure host arithmetic, no kernels."""
import numpy as np
import pytest

import ref_render as RR
from g2pc import tiles


def _layout_leaves(W, H, mt):
    """Leaves (x0, y0, w, h) in FIFO order as the renderer reaches them: the layout's own leaves by tile_seq, then the children
    of the nodes still too large, run by run and level by level (GaussHipRenderer._render_tree without Gaussian counts)."""
    lay = tiles.python_quadtree_layout(W, H, mt, 2)
    nx = lay["nx"]
    force = lay.get("tile_force", np.zeros((lay["nx"] * lay["ny"],), np.uint8))
    out, parents = [], []
    for t in lay["seq_tile"]:
        node = (int(lay["xs"][t % nx]), int(lay["ys"][t // nx]), int(lay["ws"][t % nx]), int(lay["hs"][t // nx]))
        if force[t]:
            parents.append(node + ((int(lay["tile_seq"][t]),),))
        else:
            out.append(node)
    while parents:
        nxt = []
        for host, children in tiles.child_layout(W, H, parents, 2):
            for (t, x0, y0, w, h, order) in children:
                if w > mt or h > mt:
                    nxt.append((x0, y0, w, h, order))
                else:
                    out.append((x0, y0, w, h))
        parents = nxt
    return out, bool(force.any())


SIZES = [(1280, 720, 60), (1920, 1080, 60), (961, 540, 60), (975, 720, 60), (1936, 1089, 60), (3841, 2161, 60), (121, 90, 60),
         (59, 56, 14), (241, 130, 30), (66, 97, 12), (640, 480, 60), (1957, 1091, 60), (333, 187, 25)]


@pytest.mark.parametrize("W,H,mt", SIZES)
def test_layout_leaves_are_the_queues_leaves(W, H, mt):
    want = [(x0, y0, w, h) for (kind, x0, y0, w, h, _) in
            RR.quadtree_leaves(W, H, lambda *a: np.ones((1,), bool), mt, 10 ** 9) if kind == "leaf"]
    got, forced = _layout_leaves(W, H, mt)
    assert got == want
    assert forced == ((W, H, mt) in {(961, 540, 60), (975, 720, 60), (1936, 1089, 60), (3841, 2161, 60), (121, 90, 60),
                                     (59, 56, 14), (241, 130, 30), (66, 97, 12)})


def test_random_sizes_against_the_queue():
    rng = np.random.default_rng(7)
    n_forced = n_refused = 0
    for _ in range(400):
        W, H = int(rng.integers(64, 2600)), int(rng.integers(48, 1600))
        want = [(x0, y0, w, h) for (kind, x0, y0, w, h, _) in
                RR.quadtree_leaves(W, H, lambda *a: np.ones((1,), bool), 60, 10 ** 9) if kind == "leaf"]
        try:
            got, forced = _layout_leaves(W, H, 60)
        except NotImplementedError:
            n_refused += 1          # a node narrower than two pixels before both sides fit the limit (extreme aspect ratios)
            continue
        n_forced += forced
        assert got == want, (W, H)
    assert n_forced >= 5 and n_refused <= 30, (n_forced, n_refused)
