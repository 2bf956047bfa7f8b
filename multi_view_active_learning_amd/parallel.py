"""Multi-GPU glue: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over
xGMI on ROCm) on the GPU box, "gloo" in the CPU tests.

The reference shards frames with a DistributedSampler and then issues 8 tiny all_gathers
PER SAMPLE (strategy.py:1106-1114).  Frames are independent through the whole hot path, so
here a scoring pass ends with TWO collectives -- one exchange of three integers per rank (row count, batch count, error
flag) and one padded all_gather of bytes that carries the packed result table together with the rank's batch-size list
-- and the core-set pass gathers the features once before the (replicated, communication-free) greedy loop
(SURVEY 8(e))."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# ---- overlap of a batch's post-network stage with the next batch's network ---------------------------------------------
# Per batch the path is network -> decode / scoring / RANSAC-DLT -> a row of the result table; the rows are only read when the
# pass ends (strategy.py:1004-1147 appends per sample and gathers at the end).  The post-network kernels are latency-bound (a
# few waves per CU: ransac_dlt 175 us, the MPE / BSB statistics 160 - 200 us per batch) and sit serially behind the network on
# one stream; on a side stream they run beside the NEXT batch's convolutions.  Tensors created inside the context belong to
# the side stream; tensors handed in from the main stream are marked with record_stream.
_POST_STREAMS = {}


class PostStream:
    """``with post.batch(heatmaps, keys, dp): table = score(...)`` per batch -- pass EVERY device tensor the body reads (a dict
    counts with its values) --, ``post.join()`` before the results are read on the caller's stream.  A no-op for host tensors
    and for PostStream(enabled=False) (bench.py --no-overlap: the A/B of the overlap)."""

    def __init__(self, enabled=None):
        self.enabled = True if enabled is None else bool(enabled)
        self.side = None

    class _Ctx:
        def __init__(self, outer, inputs):
            # every device tensor the body reads that was produced on the caller's stream: dicts (a batch ``dp``) count with their values.
            # Each is record_stream()ed, so rebinding it on the next loop iteration cannot hand its block back to the caller's stream
            # while the side stream still reads it (a device-resident loader / collate)
            flat = []
            for t in inputs:
                flat.extend(t.values() if isinstance(t, dict) else (t,))
            self.outer, self.inputs, self.cm = outer, flat, None

        def __enter__(self):
            o = self.outer
            t0 = next((t for t in self.inputs if torch.is_tensor(t) and t.is_cuda), None)
            if not o.enabled or t0 is None:
                return self
            key = t0.device.index
            if key not in _POST_STREAMS:
                _POST_STREAMS[key] = torch.cuda.Stream(device=t0.device)
            o.side = _POST_STREAMS[key]
            o.side.wait_stream(torch.cuda.current_stream(t0.device))  # the batch's heat-maps (and arg-max keys) are complete
            for t in self.inputs:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(o.side)
            self.cm = torch.cuda.stream(o.side)
            self.cm.__enter__()
            return self

        def __exit__(self, *exc):
            if self.cm is not None:
                self.cm.__exit__(*exc)
            return False

    def batch(self, *inputs):
        return PostStream._Ctx(self, inputs)

    def join(self):
        """The caller's current stream waits for everything queued on the side stream."""
        if self.side is not None:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int):
    """Contiguous block of ceil(n / G) frames per rank (last rank may be short)."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _collectives_on():
    """Collectives run when more than one rank exists -- or, with MVAL_DIST_NO_SHORTCUT=1, whenever a process group is
    initialised (world 1 included: the RCCL path of a single-GPU box, tests/test_gpu_distributed.py)."""
    import os

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MVAL_DIST_NO_SHORTCUT") == "1"


# Seconds spent inside the collectives of the passes since the last reset_timers() (host wall clock around each
# _gather_packed call, device synchronised on both sides when timing is on): bench.py's N > 1 line reports them so that a
# multi-GPU number can be attributed (per-rank compute vs gather vs selection).
_TIMERS = {"gather_s": 0.0, "gather_calls": 0, "on": False}


def reset_timers(on: bool = True):
    _TIMERS.update(gather_s=0.0, gather_calls=0, on=bool(on))


def timers():
    return dict(_TIMERS)


def _gather_packed(t: torch.Tensor, extra=None, flag: int = 0):
    """The two collectives of a pass: ONE ``all_gather_into_tensor`` of three int64 per rank ([rows of t, len(extra), flag])
    and ONE ``all_gather_into_tensor`` of bytes, padded to the largest rank, carrying t's rows AND the int64 list ``extra``
    (a rank's batch sizes) behind them.  One host copy (the stacked headers); the per-rank tables are VIEWS of the
    gathered buffer.  Returns (per-rank tensors shaped like t, per-rank int64 lists, per-rank flags)."""
    import time

    timed = _TIMERS["on"]
    if timed:
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        t0 = time.perf_counter()
    ws = dist.get_world_size()
    dev = t.device
    t = t.contiguous()
    ex = torch.as_tensor(list(extra) if extra is not None else [], dtype=torch.int64, device=dev).reshape(-1)
    head = torch.tensor([t.shape[0], ex.shape[0], int(flag)], dtype=torch.int64, device=dev)
    heads_t = torch.empty(ws * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads_t, head)
    heads = heads_t.cpu().reshape(ws, 3).tolist()  # the pass's one device -> host copy before the data gather
    row_bytes = t.element_size() * int(np.prod(t.shape[1:], dtype=np.int64)) if t.dim() > 1 else t.element_size()
    need = [h[0] * row_bytes + h[1] * 8 for h in heads]
    cap = (max(max(need), 8) + 7) // 8 * 8  # (8-byte slots: every rank's table and int64 list stay aligned for the views below)
    pad = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine = torch.cat([t.reshape(-1).view(torch.uint8), ex.view(torch.uint8)])
    pad[: mine.shape[0]] = mine
    buf = torch.empty(ws * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(buf, pad)
    tabs = [buf[r * cap : r * cap + heads[r][0] * row_bytes].view(t.dtype).reshape((heads[r][0],) + tuple(t.shape[1:])) for r in range(ws)]
    lists = [[] for _ in range(ws)]
    if any(h[1] for h in heads):  # every rank's batch-size list: one device-side concatenation, one host copy
        ex_host = torch.cat([buf[r * cap + heads[r][0] * row_bytes : r * cap + need[r]] for r in range(ws)]).cpu().view(torch.int64).tolist()
        pos = 0
        for r in range(ws):
            lists[r] = ex_host[pos : pos + heads[r][1]]
            pos += heads[r][1]
    if timed:
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        _TIMERS["gather_s"] += time.perf_counter() - t0
        _TIMERS["gather_calls"] += 1
    return tabs, lists, [h[2] for h in heads]


def _gather_ragged(t: torch.Tensor):
    """Per-rank tensors of a ragged first dimension (one size exchange + one padded data gather)."""
    return _gather_packed(t)[0]


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
    if not _collectives_on():
        return t
    per_rank, sizes, _ = _gather_packed(t, batch_sizes)  # one size exchange + one data gather
    base = np.concatenate([[0], np.cumsum([x.shape[0] for x in per_rank])])
    idx = torch.tensor([int(base[r]) + row for r, row in reference_gather_order(sizes)], dtype=torch.int64, device=t.device)
    return torch.cat(per_rank, dim=0).index_select(0, idx)


def all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """Concatenate a per-rank tensor along dim 0 in rank order (ragged first dim allowed)."""
    if not _collectives_on():
        return t
    return torch.cat(_gather_ragged(t), dim=0)


def gather_tables(local: torch.Tensor, batch_sizes=None, error_flag: int = 0):
    """-> list over ranks of host float64 arrays (each rank's packed scoring table).  With ``batch_sizes`` (this
    rank's list of batch sizes) also every rank's list, so that all ranks can rebuild the reference's
    (batch, sample, rank) order even when shards are ragged: (tables, sizes_per_rank).  One size exchange + one data
    gather per pass; ``error_flag`` != 0 on any rank comes back as RuntimeError on EVERY rank after the collectives
    (a rank that raised before them would leave the others waiting)."""
    if not _collectives_on():
        tabs = [local.cpu().numpy()]
        return tabs if batch_sizes is None else (tabs, [list(batch_sizes)])
    tabs, sizes, flags = _gather_packed(local, batch_sizes, error_flag)
    if any(flags) and not error_flag:
        raise RuntimeError(f"scoring pass failed on rank(s) {[r for r, f in enumerate(flags) if f]}")
    tabs = [x.cpu().numpy() for x in tabs]
    return tabs if batch_sizes is None else (tabs, sizes)
