"""Frame sharding for multi-GPU inference: one process per GPU, frames are independent units.

The reference runs one frame at a time on one GPU (rt_test.py:181-202) and its only parallelism is
data-parallel (main.py:240-245).  For inference nothing is exchanged on the data path: rank r takes frames
r, r+world, r+2*world, ... and the compact per-frame results (<= 80 KB/frame) are gathered to
rank 0 with one all_gather_object over torch.distributed -- RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence


def frame_shard(n_frames: int, rank: int, world: int) -> List[int]:
    """Indices of the frames rank `rank` owns (round-robin, so a stream stays balanced)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_frames, world))


def merge_shards(per_rank: Sequence[Sequence], n_frames: int) -> list:
    """Inverse of frame_shard: per_rank[r][k] is the result of frame r + k*world."""
    world = len(per_rank)
    out = [None] * n_frames
    for r, items in enumerate(per_rank):
        idx = frame_shard(n_frames, r, world)
        if len(items) != len(idx):
            raise ValueError(f"rank {r} returned {len(items)} results for {len(idx)} frames")
        for i, it in zip(idx, items):
            out[i] = it
    return out


def gather_results(local_results: list, n_frames: int, group=None):
    """All ranks call this; returns the frame-ordered list on every rank (small objects only)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return merge_shards([local_results], n_frames)
    world = dist.get_world_size(group)
    buf = [None] * world
    dist.all_gather_object(buf, local_results, group=group)
    return merge_shards(buf, n_frames)
