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


def reference_gather_order(sizes_per_rank):
    """The order in which the reference's per-sample all_gathers append results (strategy.py:1024,1036,1106-1145 and
    :600-636): for batch: for sample: for rank.  sizes_per_rank: one list of batch sizes per rank -> list of
    (rank, row of that rank's table).  Ragged ranks (a short last shard or batch) are walked over the longest
    structure, skipping what a rank does not have, so every rank derives the same order."""
    sizes = [[int(x) for x in b] for b in sizes_per_rank]
    offsets = [np.concatenate([[0], np.cumsum(b)]).astype(np.int64) for b in sizes]
    order = []
    for bi in range(max((len(b) for b in sizes), default=0)):
        for si in range(max((b[bi] for b in sizes if bi < len(b)), default=0)):
            for r, (b, off) in enumerate(zip(sizes, offsets)):
                if bi < len(b) and si < b[bi]:
                    order.append((r, int(off[bi]) + si))
    return order


def all_gather_reference_order(t: torch.Tensor, batch_sizes) -> torch.Tensor:
    """Rows of every rank's ``t`` (this rank's per-sample results, batch after batch) in the reference's gather
    order -- with DistributedSampler's strided shards that IS the dataset order, so order-sensitive float32
    reductions over the gathered rows (compute_mkpe) see the samples as the reference does."""
    if world()[1] == 1:
        return t
    per_rank = _gather_ragged(t)
    mine = torch.tensor(list(batch_sizes), dtype=torch.int64, device=t.device).reshape(-1)
    sizes = [x.cpu().tolist() for x in _gather_ragged(mine)]
    base = np.concatenate([[0], np.cumsum([x.shape[0] for x in per_rank])])
    idx = torch.tensor([int(base[r]) + row for r, row in reference_gather_order(sizes)], dtype=torch.int64, device=t.device)
    return torch.cat(per_rank, dim=0).index_select(0, idx)


def all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """Concatenate a per-rank tensor along dim 0 in rank order (ragged first dim allowed)."""
    if world()[1] == 1:
        return t
    return torch.cat(_gather_ragged(t), dim=0)


def gather_tables(local: torch.Tensor, batch_sizes=None):
    """-> list over ranks of host float64 arrays (each rank's packed scoring table).  With ``batch_sizes`` (this
    rank's list of batch sizes) also every rank's list, so that all ranks can rebuild the reference's
    (batch, sample, rank) order even when shards are ragged: (tables, sizes_per_rank)."""
    if world()[1] == 1:
        tabs = [local.cpu().numpy()]
        return tabs if batch_sizes is None else (tabs, [list(batch_sizes)])
    tabs = [x.cpu().numpy() for x in _gather_ragged(local)]
    if batch_sizes is None:
        return tabs
    mine = torch.tensor(list(batch_sizes), dtype=torch.int64, device=local.device).reshape(-1)
    return tabs, [x.cpu().tolist() for x in _gather_ragged(mine)]
