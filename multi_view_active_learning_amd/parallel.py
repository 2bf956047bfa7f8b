"""Multi-GPU glue: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over
xGMI on ROCm) on the GPU box, "gloo" in the CPU tests.

The reference shards frames with a DistributedSampler and then issues 8 tiny all_gathers
PER SAMPLE (strategy.py:1106-1114).  Frames are independent through the whole hot path, so
here ranks exchange exactly one packed table per scoring pass, and core-set features once
before the (replicated, communication-free) greedy loop."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int):
    """Contiguous block of ceil(n / G) frames per rank (last rank may be short)."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _gather_ragged(t: torch.Tensor):
    """ONE size exchange + ONE padded all_gather; returns the per-rank tensors."""
    rank, ws = world()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = torch.zeros((max(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad.contiguous())
    return [bufs[r][: sizes[r]] for r in range(ws)]


def all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """Concatenate a per-rank tensor along dim 0 in rank order (ragged first dim allowed)."""
    if world()[1] == 1:
        return t
    return torch.cat(_gather_ragged(t), dim=0)


def gather_tables(local: torch.Tensor):
    """-> list over ranks of host float64 arrays (each rank's packed scoring table)."""
    if world()[1] == 1:
        return [local.cpu().numpy()]
    return [x.cpu().numpy() for x in _gather_ragged(local)]
