"""Multi-GPU decomposition of the path: records are independent, so ranks take disjoint records and the
only thing that ever crosses GPUs is the per-verdict counter vector (one all-reduce, 320 bytes).

Two placements are provided:
  * flow hash     -- rank = flow_hash5(src, dst, proto, sport, dport) % world, so all frames of a flow
                     stay on one GPU: what bench.py uses for packets (weak scaling: the global batch
                     grows with the world size);
  * index ranges  -- rank r owns trace indices [r*n, (r+1)*n): bare headers, which carry no ports.
The reference has nothing distributed; these helpers only wrap torch.distributed (RCCL on GPUs, gloo in
the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

V_COUNT = 40


def rank_index_range(rank: int, world: int, frames_per_gpu: int) -> Tuple[int, int]:
    """(first trace index, record count) of a rank under weak scaling."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank * frames_per_gpu, frames_per_gpu


def shard_of_hash(hashes: np.ndarray, world: int) -> np.ndarray:
    """GPU index of every record from its 32-bit flow hash."""
    return (np.asarray(hashes, dtype=np.uint32) % np.uint32(world)).astype(np.int32)


def shard_skew(shards: np.ndarray, world: int) -> float:
    """max / mean shard size (1.0 = perfectly even)."""
    sizes = np.bincount(shards, minlength=world).astype(np.float64)
    return float(sizes.max() / max(sizes.mean(), 1.0))


def allreduce_counts(counts: torch.Tensor, dist=None) -> torch.Tensor:
    """Sum the V_COUNT verdict counters over all ranks (in place); no-op without a process group."""
    if counts.numel() != V_COUNT or counts.dtype != torch.int64:
        raise ValueError(f"expected {V_COUNT} int64 counters")
    if dist is not None and dist.is_initialized():  # (also with one rank: the collective is the code path under test then)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def max_over_ranks(seconds: float, device, dist=None) -> float:
    """The job's time is the slowest rank's time."""
    if dist is None or not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
