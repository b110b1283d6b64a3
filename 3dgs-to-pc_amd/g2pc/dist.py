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


def gather_rows(t: Optional[torch.Tensor], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Concatenate the ranks' [m_r, k] tensors on `dst` in rank order (None elsewhere)."""
    rank, world = rank_world(group)
    if world == 1 or t is None:
        return t
    t = t.contiguous()
    counts = torch.zeros((world,), dtype=torch.int64, device=t.device)
    counts[rank] = t.shape[0]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    counts = counts.tolist()
    if rank == dst:
        out = torch.empty((int(sum(counts)),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + int(c))
        out[offs[rank]:offs[rank + 1]] = t
        ops_ = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world)
                if r != dst and counts[r] > 0]
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        return out
    if t.shape[0] > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, dst, group)]):
            req.wait()
    return None


def gather_pointcloud(cloud, dst: int = 0, group=None):
    """PointCloudData of every rank's shard -> the whole cloud on `dst` (None on the other ranks)."""
    pts = gather_rows(cloud.points, dst, group)
    cols = gather_rows(cloud.colours, dst, group)
    nrm = gather_rows(cloud.normals, dst, group) if cloud.normals is not None else None
    if rank_world(group)[0] != dst:
        return None
    return type(cloud)(points=pts, colours=cols, normals=nrm)
