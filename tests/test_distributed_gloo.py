"""CPU, world_size 2 over gloo: the N>1 glue (frame sharding + one size exchange and ONE packed data gather per pass) through the
real entry points ``_compute_sal_dict`` / ``evaluate_mkpe``-style gathers / ``select_al_guids``.

No GPU here, so the two device stages of a pass are stood in for by deterministic fakes
(``_compute_batch_heatmap`` and ``score_batch`` of a test subclass); everything between them and the
returned dict -- loader walk, deferred checks, table packing, the collectives, the reference's
(batch, sample, rank) order, ragged shards -- is the product code.  The same passes with the real
kernels and two ranks sharing one GPU run in tests/test_gpu_distributed.py.
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
J = 3


def _frame_row(pose, f):
    """[pose, frame, al, sal, inliers, mkpe, 3J key-points] of one frame, a pure function of (pose, frame)."""
    rng = np.random.default_rng(1000 * pose + f)
    return [pose, f, float(rng.standard_normal()), float(rng.random()), float(rng.integers(2, 5)), float(rng.random())] + \
        rng.standard_normal(3 * J).tolist()


def _loader(frames, batch):
    """Batches of a rank's frames as the strategy's data-loader dicts (only what the fakes read)."""
    out = []
    for i in range(0, len(frames), batch):
        chunk = frames[i:i + batch]
        out.append({"pose": torch.tensor([p for p, _ in chunk]), "frame_id": torch.tensor([f for _, f in chunk]),
                    "images": torch.zeros(len(chunk), 1, 3, 4, 4)})
    return out


def _strategy():
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    class FakeDeviceStages(ActiveLearningStrategy):
        @staticmethod
        def _compute_batch_heatmap(pose_estimator, data):
            return data["images"]

        def score_batch(self, heatmaps, dp):
            rows = [_frame_row(int(p), int(f)) for p, f in zip(dp["pose"].tolist(), dp["frame_id"].tolist())]
            return torch.tensor(rows, dtype=torch.float64).reshape(-1, 6 + 3 * J)

    cfg = get_default_configs()
    cfg.DATA.NUM_JOINTS = J
    return FakeDeviceStages(cfg)


def _init(rank, world, path):
    import sys

    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method="file://" + path)


ALL = [(1, f) for f in range(7)] + [(2, f) for f in range(4)]  # 11 frames of two poses


def _worker(rank, world, path, out, mode):
    _init(rank, world, path)
    from multi_view_active_learning_amd import parallel

    if mode == "strided":      # DistributedSampler: rank r sees frames r, r + world, ... (padded by wrap-around upstream)
        mine = ALL[rank::world]
    else:                      # contiguous blocks with a short last rank (parallel.shard_range)
        lo, hi = parallel.shard_range(len(ALL), rank, world)
        mine = ALL[lo:hi]
    st = _strategy()
    d = st._compute_sal_dict(_loader(mine, 2), None)
    # order-sensitive gather used by evaluate_mkpe
    local = torch.tensor([[p, f] for p, f in mine], dtype=torch.float64).reshape(-1, 2)
    sizes = [min(2, len(mine) - i) for i in range(0, len(mine), 2)]
    ordered = parallel.all_gather_reference_order(local, sizes)
    cat = parallel.all_gather_cat(local)
    picks = st.select_al_guids(d, 3)
    torch.save({"dict": d, "ordered": ordered, "cat": cat, "picks": picks}, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def _world1():
    st = _strategy()
    d = st._compute_sal_dict(_loader(ALL, 2), None)
    return d, st.select_al_guids(d, 3)


def _run(tmp_path, mode):
    sync, out = str(tmp_path / ("sync" + mode)), str(tmp_path / ("out" + mode))
    mp.spawn(_worker, args=(2, sync, out, mode), nprocs=2, join=True)
    return [torch.load(out + ".%d" % r, weights_only=False) for r in range(2)]


def test_sal_dict_two_ranks_strided_equals_one_rank(tmp_path):
    """DistributedSampler-style shards: the reference's (batch, sample, rank) gather order IS the dataset
    order, so the world-2 dicts equal the world-1 dicts, key order included, on EVERY rank; so do the picks."""
    r0, r1 = _run(tmp_path, "strided")
    want, want_picks = _world1()
    for r in (r0, r1):
        for k in ("al_metric", "sal_metric", "inlier_count", "pred_3d_keypoints", "mkpe"):
            assert list(r["dict"][k].items()) == list(want[k].items()), k
        assert r["picks"] == want_picks
        # evaluate_mkpe's gather: dataset order again
        assert r["ordered"].tolist() == [[float(p), float(f)] for p, f in ALL]
    assert r0["cat"].tolist() == [[float(p), float(f)] for p, f in ALL[0::2] + ALL[1::2]]  # rank-major concatenation


def test_sal_dict_two_ranks_ragged_blocks(tmp_path):
    """Contiguous blocks with a short last rank (6 + 5 frames, a short last batch on rank 1): both ranks must
    build the SAME dict (the round-1 code dropped the longer rank's tail on the short rank) in
    (batch, sample, rank) order, with every frame present exactly once."""
    r0, r1 = _run(tmp_path, "blocks")
    want, _ = _world1()
    keys0 = list(r0["dict"]["al_metric"])
    assert keys0 == list(r1["dict"]["al_metric"])
    assert r0["picks"] == r1["picks"]
    a, b = ALL[:6], ALL[6:]
    expect = []
    for i in range(0, 6, 2):          # batch
        for s in range(2):            # sample
            for blk in (a, b):        # rank
                if i + s < len(blk):
                    expect.append("%d-%d" % blk[i + s])
    assert keys0 == expect and sorted(keys0) == sorted(want["al_metric"])
    for k in want:
        assert {g: r0["dict"][k][g] for g in keys0} == {g: want[k][g] for g in keys0}


def test_tables_to_sal_dict_flat_sizes_mean_every_rank():
    from multi_view_active_learning_amd.strategy import tables_to_sal_dict

    tabs = [np.asarray([_frame_row(r, f) for f in range(4)]) for r in range(2)]
    d = tables_to_sal_dict(tabs, [2, 2])
    assert list(d["al_metric"]) == ["0-0", "1-0", "0-1", "1-1", "0-2", "1-2", "0-3", "1-3"]
    try:
        tables_to_sal_dict(tabs, [[2, 2], [2, 1]])
    except ValueError:
        pass
    else:
        raise AssertionError("row count / batch size mismatch must raise")


def test_shard_range_covers_everything():
    from multi_view_active_learning_amd import parallel

    for n in (0, 1, 7, 8, 50000):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_counts(rank, world, path, out):
    """One scoring pass with the collectives counted; then a pass in which rank 1's deferred check fails."""
    _init(rank, world, path)
    from multi_view_active_learning_amd import parallel

    calls = []
    real, real_list = dist.all_gather_into_tensor, dist.all_gather
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append(1), real_list(*a, **k))[1]  # (the list form would count as well)
    st = _strategy()
    mine = ALL[rank::world]
    st._compute_sal_dict(_loader(mine, 2), None)
    n_pass = len(calls)
    local = torch.tensor([[p, f] for p, f in mine], dtype=torch.float64).reshape(-1, 2)
    parallel.all_gather_reference_order(local, [min(2, len(mine) - i) for i in range(0, len(mine), 2)])
    n_eval = len(calls) - n_pass

    class Failing(type(st)):
        def _raise_deferred_errors(self):
            if rank == 1:
                raise IndexError("list index out of range")

    st2 = Failing(st.al_cfg)
    try:
        st2._compute_sal_dict(_loader(mine, 2), None)
        err = None
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    dist.all_gather_into_tensor, dist.all_gather = real, real_list
    torch.save({"n_pass": n_pass, "n_eval": n_eval, "err": err}, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_a_pass_is_two_collectives_and_errors_reach_every_rank(tmp_path):
    """One size exchange + one data gather per scoring pass (the table AND the batch-size list travel together), the
    same for evaluate_mkpe's ordered gather; a rank whose per-pass check fails still takes part in them and every rank
    raises afterwards (the failing one its own error, the others a RuntimeError) -- nobody is left waiting."""
    sync, out = str(tmp_path / "sync_c"), str(tmp_path / "out_c")
    mp.spawn(_worker_counts, args=(2, sync, out), nprocs=2, join=True)
    r0, r1 = (torch.load(out + ".%d" % r, weights_only=False) for r in range(2))
    assert r0["n_pass"] == r1["n_pass"] == 2
    assert r0["n_eval"] == r1["n_eval"] == 2
    assert r1["err"] == "IndexError" and r0["err"] == "RuntimeError"
