"""
Tile layouts of the rasteriser.

`python_quadtree_layout` replays the FIFO quad-tree of the reference's pure-torch renderer
(gauss_render.py:290-335) for an image size and tile limit.  When no tile overflows
`max_gaussians_per_tile`, the split rule (`w > max_tile_size or h > max_tile_size` halves BOTH sides,
children = ceil(size/2), second child starts at floor(ceil(size/2)) and is clipped to the IMAGE, not the
parent) produces leaves of uniform depth that are the cartesian product of two 1-D interval families, with
1-pixel overlaps where a size was odd.  The leaves' FIFO order (top-left, bottom-left, top-right,
bottom-right per split) is kept as `tile_seq`; it decides which tile wins ties of the running maximum and
which tile's colour an overlapped pixel finally shows.

(Image sizes whose border nodes fit the limit a level before the interior ones: see python_quadtree_layout, `tile_force`.)

The tree's two DATA-DEPENDENT rules are decided per camera on top of that leaf grid: `_tree_tables` describes the tree above
the leaves (interior node extents, which leaves reach beyond which ancestor) for the rasteriser's gate, which skips leaves
under empty nodes (gauss_render.py:311-314); `child_layout` builds the next level below the nodes the reference splits for
their Gaussian count (:319-335) -- used by GaussHipRenderer._render_tree.

`grid_layout` is the regular 16x16 grid of the native rasteriser (config.h:15-17, auxiliary.h:45-55).
"""
from __future__ import annotations

from math import ceil, floor
from typing import Dict

import numpy as np

SUBBLOCKS_PER_CHUNK = 2      # 8x8-pixel sub-blocks per blend wave (1, 2 or 4 pixels per lane; 2 = packed-f32 kernel)


def _chunks_of_tile(nsbx: int, nsby: int, subblocks: int):
    """chunk_pix0 entries of one tile of nsbx x nsby 8x8 sub-blocks (numbered row-major).  1 or 4 sub-blocks per chunk:
    the first sub-block of each run of consecutive ones.  2 (the packed kernel): pairs of ADJACENT sub-blocks `a | b << 16`
    (b = 0xFFFF: none) -- side by side along each row, the odd last column paired downwards -- so that a chunk is a
    compact 16x8 / 8x16 pixel rectangle and far-away Gaussians can be culled per chunk (k_blend_py_pk)."""
    if subblocks != 2:
        return list(range(0, nsbx * nsby, subblocks))
    out = []
    for r in range(nsby):
        for c in range(0, nsbx - 1, 2):
            out.append((r * nsbx + c) | ((r * nsbx + c + 1) << 16))
    if nsbx % 2:
        c = nsbx - 1
        for r in range(0, nsby - 1, 2):
            out.append((r * nsbx + c) | (((r + 1) * nsbx + c) << 16))
        if nsby % 2:
            out.append(((nsby - 1) * nsbx + c) | (0xFFFF << 16))
    return out


def _finish(xs, ws, ys, hs, seq_of, subblocks=None, tile_shard=None, blended=None) -> Dict[str, np.ndarray]:
    """seq_of: {(x0, y0): FIFO position} or an int array [ny*nx].  blended (optional): the only tiles that get blend work
    (the children of split nodes inside a level's interval product, child_layout)."""
    subblocks = SUBBLOCKS_PER_CHUNK if subblocks is None else subblocks
    nx, ny = len(xs), len(ys)
    T = nx * ny
    if isinstance(seq_of, np.ndarray):
        tile_seq = seq_of.astype(np.int32)
    else:
        tile_seq = np.zeros((T,), dtype=np.int32)
        for iy in range(ny):
            for ix in range(nx):
                tile_seq[iy * nx + ix] = seq_of[(xs[ix], ys[iy])]
    order = np.argsort(tile_seq, kind="stable")
    # compress the FIFO sequence numbers to 0..T-1 (only their order matters)
    rank = np.empty((T,), dtype=np.int32)
    rank[order] = np.arange(T, dtype=np.int32)
    seq_tile = order.astype(np.int32)
    pix = (np.asarray(hs, np.int64)[:, None] * np.asarray(ws, np.int64)[None, :]).reshape(-1)
    off = np.zeros((T + 1,), dtype=np.int64)
    off[1:] = np.cumsum(pix)
    # Blend work list.  Dispatch order = block index, and block b runs on XCD b % 8 (observed, speed only):
    #  * tiles are visited from the image centre outwards, so the long-running (dense) tiles start first and the
    #    cheap border tiles fill the tail of the launch;
    #  * tiles are taken 8 at a time and their chunks interleaved with stride 8, so all chunks of one tile land on
    #    the same XCD and re-read the tile's Gaussian list from that XCD's L2.
    W = max(x + w_ for x, w_ in zip(xs, ws))
    H = max(y + h_ for y, h_ in zip(ys, hs))
    def dist(t):
        ix, iy = t % nx, t // nx
        cx, cy = xs[ix] + ws[ix] / 2.0, ys[iy] + hs[iy] / 2.0
        return ((cx - W / 2.0) / W) ** 2 + ((cy - H / 2.0) / H) ** 2
    order_t = sorted(range(T) if blended is None else blended, key=dist)
    if tile_shard is not None:
        # multi-GPU with fewer cameras than ranks: every rank renders every camera but blends only its share of the
        # tiles (dealt out in the centre-out order, so dense and sparse tiles are spread evenly); preprocess, sort and
        # binning are replicated, the visibility exchange is the usual one (g2pc/dist.py)
        r, w = tile_shard
        order_t = order_t[r::w]
    chunk_tile, chunk_pix0 = [], []
    for g0 in range(0, len(order_t), 8):
        group = order_t[g0:g0 + 8]
        per_tile = [_chunks_of_tile((ws[t % nx] + 7) // 8, (hs[t // nx] + 7) // 8, subblocks) for t in group]
        for c in range(max(len(p) for p in per_tile)):
            for k, t in enumerate(group):
                if c < len(per_tile[k]):
                    chunk_tile.append(t)
                    chunk_pix0.append(per_tile[k][c])
    return dict(nx=nx, ny=ny, xs=np.asarray(xs, np.int32), ws=np.asarray(ws, np.int32),
                ys=np.asarray(ys, np.int32), hs=np.asarray(hs, np.int32), tile_seq=rank, seq_tile=seq_tile,
                tile_pix_off=off.astype(np.int32), chunk_tile=np.asarray(chunk_tile, np.int32),
                chunk_pix0=np.asarray(chunk_pix0, np.int64).astype(np.uint32).view(np.int32), total_pixels=int(off[-1]), chunk_subblocks=int(subblocks))


def python_quadtree_layout(width: int, height: int, max_tile_size: int = 60, subblocks=None,
                           tile_shard=None) -> Dict[str, np.ndarray]:
    """The size-driven part of the reference's queue (split while a side exceeds max_tile_size) as a tile layout.

    Every node of one depth has the same nominal size -- children are ceil(size / 2) on both sides -- and only the nodes the
    image's right / bottom border clips are smaller.  Usually all nodes reach the limit at the same depth and the leaves are
    the product of two interval families.  When the border nodes fit the limit one level EARLIER than the interior ones
    (e.g. 961 pixels: the nominal width of depth 4 is 61 > 60, the last column is clipped to 46) the layout is the product at
    that shallower depth with `tile_force[t] = 1` on the nodes that are still too large: the renderer splits those for every
    camera exactly as it splits a leaf holding too many Gaussians (level by level, child_layout), which is what the reference's
    queue does with them -- a node without Gaussians is painted and dropped BEFORE its size is looked at (gauss_render.py:311-319)."""
    queue = [([0, 0], [width, height], 0)]       # ([row, col], [w, h], depth) as in the reference
    by_depth = {}                                # depth -> [(x0, y0, w, h, is_leaf)] in FIFO order
    while queue:
        start, size, depth = queue.pop(0)
        if size[0] <= 1 or size[1] <= 1:          # gauss_render.py:301
            continue
        size = [min(size[0], width - start[1]), min(size[1], height - start[0])]
        big = size[0] > max_tile_size or size[1] > max_tile_size
        by_depth.setdefault(depth, []).append((start[1], start[0], size[0], size[1], not big))
        if big:
            size = [ceil(size[0] / 2), ceil(size[1] / 2)]
            s = list(start)
            queue.append((list(s), list(size), depth + 1))
            s[0] += floor(size[1])
            queue.append((list(s), list(size), depth + 1))
            s[0] -= floor(size[1])
            s[1] += floor(size[0])
            queue.append((list(s), list(size), depth + 1))
            s[0] += floor(size[1])
            queue.append((list(s), list(size), depth + 1))
    leaf_depths = [d for d, nodes in by_depth.items() if any(n[4] for n in nodes)]
    if not leaf_depths:
        # (every branch ends in nodes narrower than two pixels, which the reference drops, gauss_render.py:301: it paints nothing)
        raise NotImplementedError("no quad-tree leaf survives for %dx%d / %d: every node gets narrower than two pixels before "
                                  "both sides fit the tile limit" % (width, height, max_tile_size))
    d0 = min(leaf_depths)
    nodes = by_depth[d0]                          # the shallowest level holding a leaf: its leaves + the nodes still too large
    xs = sorted({(n[0], n[2]) for n in nodes})
    ys = sorted({(n[1], n[3]) for n in nodes})
    if len({x for x, _ in xs}) != len(xs) or len({y for y, _ in ys}) != len(ys) or len(xs) * len(ys) != len(nodes):
        raise NotImplementedError("quad-tree level %d is not a product of intervals for %dx%d / %d (nodes narrower than two "
                                  "pixels were dropped)" % (d0, width, height, max_tile_size))
    wof, hof = dict(xs), dict(ys)
    seq_of, force_of = {}, {}
    for s, (x0, y0, w, h, leaf) in enumerate(nodes):
        if wof[x0] != w or hof[y0] != h:
            raise NotImplementedError("non-uniform quad-tree level")
        seq_of[(x0, y0)] = s
        force_of[(x0, y0)] = 0 if leaf else 1
    lay = _finish([x for x, _ in xs], [w for _, w in xs], [y for y, _ in ys], [h for _, h in ys], seq_of, subblocks,
                  tile_shard)
    lay.update(_tree_tables(width, height, lay))
    if len(nodes) > 1 and (min(w for _, w in xs) <= 1 or min(h for _, h in ys) <= 1):
        # A leaf the image border clips to ONE pixel can have no member (the reference's test is strict on inclusive pixel
        # bounds, gauss_render.py:306-309) -- but a Gaussian whose clipped rectangle ends in that pixel IS a member of the
        # ancestors.  The rasteriser's gate derives "this ancestor holds no Gaussian" (:311-314) from the members of the leaves
        # and would call such an ancestor empty (tools/experiments/quadtree_fuzz.py 101, case 52: 272 x 48 at max_tile_size 14,
        # leaves of 9 x 2 and 9 x 1 pixels).  Only trees with leaves one or two pixels thin have such leaves; the renderer then
        # takes the "which nodes are empty" decision on the host, with the reference's own count (GaussHipRenderer._static_plan).
        lay["thin_leaves"] = True
    force = np.array([force_of[(x, y)] for (y, _) in ys for (x, _) in xs], dtype=np.uint8)
    if force.any():
        lay["tile_force"] = force                 # [ny*nx]: 1 = larger than max_tile_size, always split (G2pcTileLayout.tile_force)
    return lay


def split_interval(x0: int, w: int, limit: int):
    """The two children of a node's interval along one axis (gauss_render.py:321-334 and :304-305): both ceil(w/2) wide, the
    second starting right after the first and clipped to the IMAGE -- it reaches one pixel beyond an odd-sized parent."""
    c = ceil(w / 2)
    return (x0, c), (x0 + c, min(c, limit - (x0 + c)))


def _axis_levels(length: int, depth: int):
    """Intervals (start, size) of every level 0 .. depth of the uniform tree along one axis."""
    levels = [[(0, length)]]
    for _ in range(depth):
        levels.append([c for iv in levels[-1] for c in split_interval(iv[0], iv[1], length)])
    return levels


def _tree_tables(width: int, height: int, lay) -> Dict[str, np.ndarray]:
    """Quad-tree information of a leaf layout for the rasteriser's gate (G2pcTileLayout.depth / inner_x / inner_y / tile_stick):
    the pixel extents of the interior nodes per axis and, per leaf, the levels whose ancestor it reaches beyond."""
    nx, ny = lay["nx"], lay["ny"]
    depth = int(round(np.log2(nx))) if nx > 1 else 0
    none = dict(depth=0, inner_x=np.zeros((1, 2), np.int32), inner_y=np.zeros((1, 2), np.int32), tile_stick=np.zeros((nx * ny,), np.int32))
    if nx != ny or (1 << depth) != nx or depth == 0:
        return none
    lx, ly = _axis_levels(width, depth), _axis_levels(height, depth)
    if [tuple(v) for v in lx[depth]] != list(zip(lay["xs"].tolist(), lay["ws"].tolist())) or \
            [tuple(v) for v in ly[depth]] != list(zip(lay["ys"].tolist(), lay["hs"].tolist())):
        return none
    inner_x = np.array([(x0, x0 + w - 1) for k in range(depth) for (x0, w) in lx[k]], np.int32)
    inner_y = np.array([(y0, y0 + h - 1) for k in range(depth) for (y0, h) in ly[k]], np.int32)
    x1 = lay["xs"] + lay["ws"] - 1
    y1 = lay["ys"] + lay["hs"] - 1
    sx = np.zeros((nx,), np.int32)
    sy = np.zeros((ny,), np.int32)
    for k in range(depth):
        base, sh = (1 << k) - 1, depth - k
        sx |= ((x1 > inner_x[base + (np.arange(nx) >> sh), 1]).astype(np.int32) << k)
        sy |= ((y1 > inner_y[base + (np.arange(ny) >> sh), 1]).astype(np.int32) << k)
    stick = (sy[:, None] | sx[None, :]).reshape(-1).astype(np.int32)
    return dict(depth=depth, inner_x=inner_x, inner_y=inner_y, tile_stick=stick)


def child_layout(width: int, height: int, parents, subblocks=None):
    """The next quad-tree level below a set of split nodes, as tile layouts.

    parents: list of (x0, y0, w, h, order) -- the nodes the reference splits (gauss_render.py:319-335), `order` any sortable
    key that reproduces their FIFO order.  Returns a list of RUNS [(layout, children), ...] in FIFO order (usually one): a
    run's layout is the product of the distinct child column intervals with the distinct child row intervals (the children
    of a node no larger than 2 pixels a side are dropped, :301), and children[i] = (tile index, x0, y0, w, h, order + (c,))
    lists the tiles that ARE children of a parent, in FIFO order (c = 0 top-left, 1 bottom-left, 2 top-right, 3 bottom-right,
    :325-333).  The other tiles of the product are not part of the tree: they get no blend work and the caller masks them out
    of the gate and the image (tile_mask); a caller that shards the tiles over ranks deals the children out itself
    (GaussHipRenderer._render_tree).

    Why runs: nodes of one level that descend from ancestors of different sizes (a 13-pixel node split for its size beside a
    12-pixel leaf split for its Gaussian count) overlap by more than the usual pixel, and their children may START at the same
    column / row with DIFFERENT sizes -- which no product of intervals holds.  The parents are therefore cut, in FIFO order,
    into maximal runs whose children are compatible; the caller renders the runs one after the other, which is the reference's
    own order (sequence numbers and painting order across runs stay FIFO).  [] = every child is dropped."""
    runs = []
    xi, yi, kids = {}, {}, []

    def close():
        nonlocal xi, yi, kids
        if kids:
            xs, ys = sorted(xi), sorted(yi)
            colx, rowy = {x: i for i, x in enumerate(xs)}, {y: i for i, y in enumerate(ys)}
            children = [(rowy[ky] * len(xs) + colx[kx], kx, ky, kw, kh, order) for (kx, ky, kw, kh, order) in kids]
            seq = np.full((len(xs) * len(ys),), len(kids), dtype=np.int32)  # tiles outside the tree: after every child (never blended)
            seq[[c[0] for c in children]] = np.arange(len(kids), dtype=np.int32)
            lay = _finish(xs, [xi[x] for x in xs], ys, [yi[y] for y in ys], seq, subblocks, None, blended=[c[0] for c in children])
            runs.append((lay, children))
        xi, yi, kids = {}, {}, []

    for (x0, y0, w, h, order) in sorted(parents, key=lambda p: p[4]):
        if ceil(w / 2) <= 1 or ceil(h / 2) <= 1:
            continue        # the reference drops a node by its size BEFORE clipping it to the image (:301, :304-305): all four go
        cx = split_interval(x0, w, width)
        cy = split_interval(y0, h, height)
        # (a child the image border clips to ONE pixel stays in the queue: nothing can be a member of it, so the reference
        # paints it with the background over whatever an earlier node left there)
        four = [(cx[a], cy[b]) for (a, b) in ((0, 0), (0, 1), (1, 0), (1, 1))]
        if any(xi.get(kx, kw) != kw or yi.get(ky, kh) != kh for ((kx, kw), (ky, kh)) in four):
            close()         # a child of this parent starts where another one of another size does: next run
        for c, ((kx, kw), (ky, kh)) in enumerate(four):
            xi[kx], yi[ky] = kw, kh
            kids.append((kx, ky, kw, kh, tuple(order) + (c,)))
    close()
    return runs


def grid_layout(width: int, height: int, block: int = 16) -> Dict[str, np.ndarray]:
    xs = list(range(0, width, block))
    ys = list(range(0, height, block))
    ws = [min(block, width - x) for x in xs]
    hs = [min(block, height - y) for y in ys]
    seq_of = {(x, y): iy * len(xs) + ix for iy, y in enumerate(ys) for ix, x in enumerate(xs)}
    return _finish(xs, ws, ys, hs, seq_of)
