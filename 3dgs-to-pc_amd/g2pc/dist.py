"""
Multi-GPU layer (no counterpart in the reference, which is single-process): one process per GPU,
torch.distributed with the nccl backend (= RCCL over xGMI on MI355X; gloo on CPU for tests).

Sharding (BASELINE north_star):
  * rendering  -- alpha blending needs every Gaussian in front of a pixel, so the blend work is split by CAMERA:
                  rank r renders cameras r, r+W, ... against the replicated read-only scene and keeps the running
                  per-Gaussian state; ONE exchange after the camera loop combines it
                  (GaussHipRenderer.all_reduce_visibility: all-reduce MAX of the packed 64-bit
                  (contribution, ~global order) keys -- exact, order-free -- then all-reduce SUM of the winners'
                  colours, one non-zero term per Gaussian).  Payload 20 B per Gaussian.
  * sampling   -- by GAUSSIAN INDEX: every rank derives the identical global allocation (sizes, points per Gaussian,
                  bins; N-sized, microseconds) and samples only its contiguous index range; noise is keyed by the
                  global Gaussian index, so the union over ranks is exactly the single-GPU cloud.
  * gather     -- point-to-point sends of the shards to one rank (36 B per point), rank-major order.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def rank_world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def gather_rows_many(tensors, dst: int = 0, group=None):
    """Concatenate, for each tensor of the list, the ranks' [m_r, ...] pieces on `dst` in rank order (None elsewhere).
    All tensors of one rank share m_r: ONE count exchange, then ONE batch of point-to-point transfers (RCCL groups
    them, so the seven incoming xGMI links of `dst` are driven concurrently)."""
    rank, world = rank_world(group)
    present = [t for t in tensors if t is not None]
    if world == 1 or not present:
        return list(tensors)
    tensors = [t.contiguous() if t is not None else None for t in tensors]
    m = present[0].shape[0]
    counts = torch.zeros((world,), dtype=torch.int64, device=present[0].device)
    counts[rank] = m
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    counts = counts.tolist()
    if rank != dst:
        sends = [dist.P2POp(dist.isend, t, dst, group) for t in tensors if t is not None and m > 0]
        if sends:
            for req in dist.batch_isend_irecv(sends):
                req.wait()
        return [None] * len(tensors)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + int(c))
    outs, recvs = [], []
    for t in tensors:
        if t is None:
            outs.append(None)
            continue
        out = torch.empty((offs[-1],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[offs[rank]:offs[rank + 1]] = t
        outs.append(out)
    for r in range(world):                      # rank-major so the order matches the senders' posting order
        if r == dst or counts[r] == 0:
            continue
        recvs += [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for out in outs if out is not None]
    if recvs:
        for req in dist.batch_isend_irecv(recvs):
            req.wait()
    return outs


def gather_rows(t: Optional[torch.Tensor], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Concatenate the ranks' [m_r, k] tensors on `dst` in rank order (None elsewhere)."""
    return gather_rows_many([t], dst, group)[0]


def gather_pointcloud(cloud, dst: int = 0, group=None):
    """PointCloudData of every rank's shard -> the whole cloud on `dst` (None on the other ranks)."""
    pts, cols, nrm = gather_rows_many([cloud.points, cloud.colours, cloud.normals], dst, group)
    if rank_world(group)[0] != dst:
        return None
    return type(cloud)(points=pts, colours=cols, normals=nrm)
