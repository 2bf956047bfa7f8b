"""CPU, world_size 2 over gloo: the N>1 glue (frame sharding + ONE packed gather per pass)."""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, path, out):
    import sys

    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method="file://" + path)
    from multi_view_active_learning_amd import parallel
    from multi_view_active_learning_amd.strategy import tables_to_sal_dict

    n, j = 7, 3
    lo, hi = parallel.shard_range(n, rank, world)
    # packed table of this rank's frames: [pose, frame, al, sal, inl, mkpe, 3J kp]
    rows = []
    for f in range(lo, hi):
        rows.append([rank, f, float(f) * 0.5, 1.0, 4, 2.0] + [float(f)] * (3 * j))
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 6 + 3 * j)
    per_rank = parallel.gather_tables(local)
    sizes = [2] * ((hi - lo) // 2) + ([1] if (hi - lo) % 2 else [])
    d = tables_to_sal_dict(per_rank, [2, 2])  # rank 0's batch structure (4 frames)
    cat = parallel.all_gather_cat(local)
    if rank == 0:
        torch.save({"keys": list(d["al_metric"]), "cat": cat, "sizes": [t.shape[0] for t in per_rank]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather(tmp_path):
    sync = str(tmp_path / "sync")
    out = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(2, sync, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["sizes"] == [4, 3]  # ceil(7/2) frames on rank 0, the rest on rank 1
    # (batch, sample, rank) order with a ragged last rank
    assert r["keys"] == ["0-0", "1-4", "0-1", "1-5", "0-2", "1-6", "0-3"]
    assert r["cat"].shape == (7, 6 + 9) and r["cat"][:, 1].tolist() == [0, 1, 2, 3, 4, 5, 6]


def test_shard_range_covers_everything():
    from multi_view_active_learning_amd import parallel

    for n in (0, 1, 7, 8, 50000):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
