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


def _finish(xs, ws, ys, hs, seq_of, subblocks=None, tile_shard=None) -> Dict[str, np.ndarray]:
    subblocks = SUBBLOCKS_PER_CHUNK if subblocks is None else subblocks
    nx, ny = len(xs), len(ys)
    T = nx * ny
    tile_seq = np.zeros((T,), dtype=np.int32)
    for iy in range(ny):
        for ix in range(nx):
            tile_seq[iy * nx + ix] = seq_of[(xs[ix], ys[iy])]
    order = np.argsort(tile_seq, kind="stable")
    # compress the FIFO sequence numbers to 0..T-1 (only their order matters)
    rank = np.empty((T,), dtype=np.int32)
    rank[order] = np.arange(T, dtype=np.int32)
    seq_tile = order.astype(np.int32)
    pix = np.array([ws[t % nx] * hs[t // nx] for t in range(T)], dtype=np.int64)
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
    order_t = sorted(range(T), key=dist)
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
    queue = [([0, 0], [width, height])]          # ([row, col], [w, h]) as in the reference
    leaves = []
    while queue:
        start, size = queue.pop(0)
        if size[0] <= 1 or size[1] <= 1:          # gauss_render.py:301
            continue
        size = [min(size[0], width - start[1]), min(size[1], height - start[0])]
        if size[0] > max_tile_size or size[1] > max_tile_size:
            size = [ceil(size[0] / 2), ceil(size[1] / 2)]
            s = list(start)
            queue.append((list(s), list(size)))
            s[0] += floor(size[1])
            queue.append((list(s), list(size)))
            s[0] -= floor(size[1])
            s[1] += floor(size[0])
            queue.append((list(s), list(size)))
            s[0] += floor(size[1])
            queue.append((list(s), list(size)))
            continue
        leaves.append((start[1], start[0], size[0], size[1]))       # x0, y0, w, h in FIFO order
    xs = sorted({(l[0], l[2]) for l in leaves})
    ys = sorted({(l[1], l[3]) for l in leaves})
    if len({x for x, _ in xs}) != len(xs) or len({y for y, _ in ys}) != len(ys) or len(xs) * len(ys) != len(leaves):
        raise NotImplementedError("quad-tree leaves are not a product of intervals for %dx%d / %d" % (width, height, max_tile_size))
    wof, hof = dict(xs), dict(ys)
    seq_of = {}
    for s, (x0, y0, w, h) in enumerate(leaves):
        if wof[x0] != w or hof[y0] != h:
            raise NotImplementedError("non-uniform quad-tree leaves")
        seq_of[(x0, y0)] = s
    return _finish([x for x, _ in xs], [w for _, w in xs], [y for y, _ in ys], [h for _, h in ys], seq_of, subblocks,
                   tile_shard)


def grid_layout(width: int, height: int, block: int = 16) -> Dict[str, np.ndarray]:
    xs = list(range(0, width, block))
    ys = list(range(0, height, block))
    ws = [min(block, width - x) for x in xs]
    hs = [min(block, height - y) for y in ys]
    seq_of = {(x, y): iy * len(xs) + ix for iy, y in enumerate(ys) for ix, x in enumerate(xs)}
    return _finish(xs, ws, ys, hs, seq_of)
