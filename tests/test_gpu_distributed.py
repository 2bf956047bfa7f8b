"""World size 2 through the REAL device path: two processes share cuda:0 and talk over gloo (RCCL refuses two
ranks on one device; the collectives' payloads are the same tensors either way).  Pool scoring
(``_compute_sal_dict`` for an entropy strategy and for CORESET), selection (``select_al_guids``: nlargest /
k-center on device) and evaluation (``evaluate_all``) on DistributedSampler-style strided shards must give
EXACTLY what one rank computes over the whole loader: per-frame results do not depend on the batch they sit
in, and the gathers restore the reference's (batch, sample, rank) order (strategy.py:1106-1145, 600-636)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("al_metric", "sal_metric", "inlier_count", "pred_3d_keypoints", "mkpe")


def _frames(c):
    """The case's loader flattened to per-frame records (dict of arrays without the batch axis, heat-maps (V,J,h,w))."""
    loader, hms = cases.build_sal_loader(c)
    out = []
    for dp, hm in zip(loader, hms):
        b = dp["pose"].shape[0]
        hm = hm.reshape((b, -1) + hm.shape[1:])
        for i in range(b):
            out.append(({k: v[i] for k, v in dp.items()}, hm[i]))
    return out


def _batches(frames, b):
    loader, hms = [], []
    for i in range(0, len(frames), b):
        chunk = frames[i:i + b]
        loader.append({k: torch.from_numpy(np.stack([f[0][k] for f in chunk])) for k in chunk[0][0]})
        hms.append(np.concatenate([f[1] for f in chunk]))
    return loader, hms


def _passes(c, frames, labeled):
    """sal_dict, picks and evaluation of one rank's loader (collectives inside when torch.distributed is up)."""
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    dev = torch.device("cuda:0")
    cfg = get_default_configs()
    cfg.AL.STRATEGY = c["strategy"]
    cfg.POSE_ESTIMATOR.STRIDE = c["stride"]
    st = ActiveLearningStrategy(cfg)
    out = {}
    loader, hms = _batches(frames, c["b"])
    it = iter(hms)
    out["sal"] = st._compute_sal_dict(loader, lambda images: torch.from_numpy(next(it)).to(dev))
    out["picks"] = st.select_al_guids(out["sal"], c["select"], labeled)
    it = iter(hms)
    out["eval"] = st.evaluate_all(loader, lambda images: torch.from_numpy(next(it)).to(dev))
    return out


def _labeled(c):
    """A labelled set for the core-set pass: {guid: {"3d_keypoints": (rows, J)}} as get_al_dict_for_coreset builds it."""
    rng = np.random.default_rng(9)
    return {"L-%d" % i: (rng.standard_normal((c["j"], 3)) * 250.0).tolist() for i in range(5)}


def _worker(rank, world, path, out, c):
    import sys

    for p in (REPO, os.path.join(REPO, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method="file://" + path)
    frames = _frames(c)[rank::world]  # DistributedSampler: indices[rank::world]
    res = _passes(c, frames, _labeled(c) if c["strategy"] == "CORESET" else None)
    torch.save(res, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strategy", ["MPE", "CORESET"])
def test_two_ranks_one_gpu_equal_one_rank(tmp_path, strategy):
    assert torch.cuda.is_available()
    c = dict(cases.sal_cases()["mpe"], strategy=strategy, nbatch=4, select=3)
    sync, out = str(tmp_path / "sync"), str(tmp_path / "out")
    mp.spawn(_worker, args=(2, sync, out, c), nprocs=2, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(2)]
    want = _passes(c, _frames(c), _labeled(c) if strategy == "CORESET" else None)
    assert len(want["sal"]["al_metric"]) == 8
    for g in got:
        for k in FIELDS:
            assert list(g["sal"][k]) == list(want["sal"][k]), k  # key order
            for guid, val in want["sal"][k].items():  # values exactly (mkpe is NaN where a joint is invalid, as in the reference)
                assert g["sal"][k][guid] == val or (val != val and g["sal"][k][guid] != g["sal"][k][guid]), (k, guid)
        assert g["picks"] == want["picks"]
        assert g["eval"] == want["eval"]  # float32 MKPE summed in the same sample order; PCK counts


def test_rccl_world_1_carries_the_collectives(tmp_path, monkeypatch):
    """A single-GPU box cannot run two RCCL ranks, but it can run the collectives themselves: with
    MVAL_DIST_NO_SHORTCUT=1 the world-1 short-circuits are off and every gather of a pass goes through
    torch.distributed's "nccl" backend (= RCCL) with device tensors -- ragged tables, the ordered gather, the
    core-set concatenation -- and bench.py's core-set pool pass runs end to end under it."""
    import subprocess
    import sys

    from multi_view_active_learning_amd import parallel

    assert torch.cuda.is_available()
    monkeypatch.setenv("MVAL_DIST_NO_SHORTCUT", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="file://" + str(tmp_path / "rccl"), device_id=dev)
    try:
        calls = []
        real = dist.all_gather_into_tensor  # (a pass = one header exchange + one data gather, both all_gather_into_tensor)
        monkeypatch.setattr(dist, "all_gather_into_tensor", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        monkeypatch.setattr(dist, "all_gather", lambda *a, **k: (_ for _ in ()).throw(AssertionError("the list all_gather is not used")))
        t = torch.arange(7 * 5, dtype=torch.float64, device=dev).reshape(7, 5)
        tabs, sizes = parallel.gather_tables(t, [3, 3, 1])
        assert len(calls) == 2 and sizes == [[3, 3, 1]]
        np.testing.assert_array_equal(tabs[0], t.cpu().numpy())
        o = parallel.all_gather_reference_order(t.float(), [3, 3, 1])
        assert torch.equal(o, t.float()) and len(calls) == 4
        e = parallel.all_gather_cat(torch.zeros((0, 19, 3), device=dev))
        assert e.shape == (0, 19, 3) and len(calls) == 6
    finally:
        dist.destroy_process_group()
    env = dict(os.environ, MVAL_DIST_NO_SHORTCUT="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "c5", "--pool", "16", "--steps", "1", "--no-cpu-baseline",
                        "--rccl-world-1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import json

    # (RCCL prints its version banner to stdout when the process exits: the line is the last one that is JSON)
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["pool_frames"] == 16 and line["value"] > 0
    # the N > 1 line is attributable: per-rank time of the timed region and its split into compute / collectives / selection
    at = line["attribution"]
    assert at["per_rank_s"]["max"] > 0 and at["compute_s"]["max"] > 0 and at["gather_s"] >= 0 and at["select_s"] > 0
    assert at["compute_s"]["max"] + at["gather_s"] + at["select_s"] <= at["per_rank_s"]["max"] * 1.05


def test_rccl_world_1_training_step_under_ddp():
    """bench.py --workload c3 --rccl-world-1: the training step through DistributedDataParallel (segmented backward, bucketed
    all-reduce) over the "nccl" backend (= RCCL) at world size 1 -- the single-GPU rehearsal of the multi-GPU C3 line, with
    the exposed all-reduce time (step vs the same step under no_sync) in the line."""
    import json
    import subprocess
    import sys

    import gc

    gc.collect()
    torch.cuda.empty_cache()  # (the child needs ~60 GB: give back what earlier tests of THIS process left in torch's caching allocator)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29643")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--rccl-world-1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["ms_per_step"] < 200
    at = line["attribution"]
    assert at["per_rank_s"]["max"] > 0 and "allreduce_exposed_s" in at and abs(at["allreduce_exposed_s"]) < 0.05


def _ddp_train_worker(rank, world, path, out, c, lanes):
    import sys

    for p in (REPO, os.path.join(REPO, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if lanes is not None:
        os.environ["MVAL_TRAIN_LANES"] = lanes
    from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError

    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method="file://" + path)
    x, gt, valid = cases.train_input(c)
    sd = {k: torch.from_numpy(v) for k, v in cases.model_state_dict(c).items()}
    m = cases.product_model(c)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    if rank == 1:  # broadcast_buffers / the constructor's parameter broadcast must overwrite these with rank 0's
        with torch.no_grad():
            for b in m.buffers():
                if b.dtype.is_floating_point:
                    b.add_(1.0)
    net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], broadcast_buffers=True)  # workflow.py:133-138
    sl = slice(rank, None, world)  # DistributedSampler: indices[rank::world]
    xs, gs, vs = (torch.from_numpy(a[sl]).to(dev) for a in (x, gt, valid.reshape(c["n"], -1, 1, 1)))
    loss = Pose2DMeanSquaredError().pose_2d_mse(net(xs), gs, vs)
    loss.backward()
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    with torch.no_grad():
        net(xs)  # second forward: starts with the broadcast of rank 0's buffers (as every reference step does)
    torch.cuda.synchronize()
    torch.save({"loss": float(loss), "grads": grads, "buffers": {k: b.detach().cpu().clone() for k, b in m.named_buffers()}}, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lanes", [None, "0"], ids=["lanes-default", "lanes-off"])
def test_two_rank_ddp_training_step_averages_shard_gradients(tmp_path, lanes, monkeypatch):
    """workflow.py:133-138 + strategy.py:460-487 at world size 2 (two processes on cuda:0 over gloo): one training step of the
    product model under DistributedDataParallel(broadcast_buffers=True).  Every rank's gradients == the mean of the two
    single-rank shard gradients (BatchNorm statistics stay per rank: no SyncBN in the reference), and the BatchNorm buffers
    after the next forward are rank 0's (broadcast at the start of the forward) advanced by the rank's own shard -- with the
    training lanes (branches on separate streams, join-less backward) at their default and switched off."""
    from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError

    assert torch.cuda.is_available()
    if lanes is not None:
        monkeypatch.setenv("MVAL_TRAIN_LANES", lanes)
    dev = torch.device("cuda:0")
    c = dict(arch="hrnet_w32", seed=12, n=8, h=64, w=64, j=19)
    sync, out = str(tmp_path / "sync"), str(tmp_path / "out")
    mp.spawn(_ddp_train_worker, args=(2, sync, out, c, lanes), nprocs=2, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(2)]

    x, gt, valid = cases.train_input(c)
    sd = {k: torch.from_numpy(v) for k, v in cases.model_state_dict(c).items()}
    single = []
    for r in range(2):
        m = cases.product_model(c)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        sl = slice(r, None, 2)
        xs, gs, vs = (torch.from_numpy(a[sl]).to(dev) for a in (x, gt, valid.reshape(c["n"], -1, 1, 1)))
        loss = Pose2DMeanSquaredError().pose_2d_mse(m(xs), gs, vs)
        loss.backward()
        single.append(dict(loss=float(loss), grads={k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()},
                           buffers={k: b.detach().cpu().clone() for k, b in m.named_buffers()}, model=m, xs=xs))
    for r in range(2):
        assert got[r]["loss"] == single[r]["loss"]  # the forward is the rank's own shard
    worst = 0.0
    for k in single[0]["grads"]:
        want = (single[0]["grads"][k] + single[1]["grads"][k]) / 2  # (x / 2 is exact: sum-then-halve == halve-then-sum)
        for r in range(2):
            g = got[r]["grads"][k]
            assert torch.equal(got[0]["grads"][k], g), k  # every rank holds the same averaged gradient
            if not torch.equal(g, want):  # <= 1 ulp where the all-reduce rounds differently from the host sum
                ulp = torch.finfo(torch.float32).eps * want.abs().clamp_min(1e-30)
                worst = max(worst, float(((g - want).abs() / ulp).max()))
    assert worst <= 1.0, worst
    # buffers after the second forward: rank 0's buffers after step 1, advanced by one train-mode forward of the rank's own shard
    for r in range(2):
        m = single[r]["model"]
        m.load_state_dict({**{k: v for k, v in m.state_dict().items()}, **{k: v.to(dev) for k, v in single[0]["buffers"].items()}}, strict=True)
        with torch.no_grad():
            m(single[r]["xs"])
        for k, b in m.named_buffers():
            assert torch.equal(b.cpu(), got[r]["buffers"][k]), (r, k)


def _bench_line(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096  # ONE line from rank 0, inside the driver's tail
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["c5", "c4"])
def test_bench_two_rank_rehearsal_pool_passes(tmp_path, workload):
    """bench.py's own N > 1 code on a one-GPU box (--shared-device: both ranks on cuda:0 over gloo): the self-launch as a child
    torchrun BEFORE any GPU call, the barriers, the rank reductions of `repeats` / the elapsed time, per-rank attribution -- and
    the picks of the two-rank pass equal the one-rank pass over the same pool content."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    common = ["--workload", workload, "--pool", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-rooflines", "--shared-device"]
    two = _bench_line(["--gpus", "2"] + common + ["--detail-out", str(tmp_path / "d2.json")])
    one = _bench_line(["--gpus", "1"] + common + ["--detail-out", str(tmp_path / "d1.json")])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["value"] > 0 and two["scaling"] == "strong"
    assert two["config"]["pool_frames"] == 64 and "REHEARSAL" in two["config"]["parallelism"]
    at = two["attribution"]
    assert len(at["per_rank_s"]["all"]) == 2 and at["compute_s"]["max"] > 0 and at["gather_s"] >= 0 and at["select_s"] >= 0
    assert two["picks_crc"] == one["picks_crc"]


def test_bench_two_rank_rehearsal_training_step(tmp_path):
    """The C3 line at two ranks on one GPU: DistributedDataParallel over gloo, `repeats` agreed by all_reduce, the exposed
    all-reduce measurement (step vs the same step under no_sync) in the line."""
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    d = _bench_line(["--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-rooflines", "--shared-device",
                     "--min-timed-seconds", "0", "--detail-out", str(tmp_path / "d.json")])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["images_per_step_per_gpu"] == 128
    at = d["attribution"]
    assert len(at["per_rank_s"]["all"]) == 2 and "allreduce_exposed_s" in at


def test_pool_pass_issues_no_host_copy_on_the_side_stream(monkeypatch):
    """The PRODUCT's pool loop (strategy.py:1004-1147 -> ActiveLearningStrategy._compute_sal_dict), fed what the reference's DataLoader
    yields -- HOST tensors -- over 64 frames of 8 views with 96 x 72 heat-maps (the C4 shapes): everything a batch's side-stream body
    (parallel.PostStream: decode + MPE + RANSAC-DLT + the table row) reads was staged on the caller's stream beforehand, so inside the
    body there is NO host -> device copy and NO device -> host read (either one blocks the host until the side stream drains and the
    next batch's network could not be enqueued meanwhile).  Timing-free: the transfers are counted, not clocked."""
    from multi_view_active_learning_amd import parallel, synth
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.strategy import ActiveLearningStrategy

    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    b, v, j, hh, wh, nb = 8, 8, 19, 96, 72, 8
    cfg = get_default_configs()
    cfg.AL.STRATEGY = "MPE"
    cfg.POSE_ESTIMATOR.STRIDE = 4
    st = ActiveLearningStrategy(cfg)
    rng = np.random.default_rng(3)
    loader = []
    for i in range(nb):  # host tensors, pageable, as a DataLoader's collate yields them (dataset/dataset.py:158-220)
        loader.append({
            "images": torch.zeros(b, v, 3, 8, 8),
            "proj_matrices": torch.from_numpy(np.stack([synth.ring_cameras(v, hh * 4, wh * 4, seed=i * b + s) for s in range(b)])),
            "joint_valid": torch.ones(b, j, dtype=torch.uint8),
            "3d_keypoints": torch.from_numpy(rng.standard_normal((b, 3, j)) * 300.0),
            "pose": torch.full((b,), 7, dtype=torch.int64),
            "frame_id": torch.arange(i * b, (i + 1) * b, dtype=torch.int64),
        })
    g = torch.Generator(device=dev).manual_seed(5)
    maps = [torch.rand(b * v, j, hh, wh, device=dev, generator=g) for _ in range(nb)]
    it = iter(maps)

    inside = {"depth": 0, "bodies": 0}
    events = []
    real_enter, real_exit = parallel.PostStream._Ctx.__enter__, parallel.PostStream._Ctx.__exit__

    def enter(self):
        r = real_enter(self)
        inside["depth"] += 1
        inside["bodies"] += 1
        return r

    def exit_(self, *exc):
        inside["depth"] -= 1
        return real_exit(self, *exc)

    monkeypatch.setattr(parallel.PostStream._Ctx, "__enter__", enter)
    monkeypatch.setattr(parallel.PostStream._Ctx, "__exit__", exit_)

    def watch(name, kind):
        real = getattr(torch.Tensor, name)

        def f(self, *a, **k):
            if inside["depth"] > 0:
                if kind == "d2h" and self.is_cuda:
                    events.append((name, tuple(self.shape)))
                if kind == "to" and not self.is_cuda:
                    tgt = [x for x in list(a) + list(k.values()) if isinstance(x, (torch.device, str)) or (torch.is_tensor(x) and x.is_cuda)]
                    if name == "cuda" or any("cuda" in str(getattr(x, "device", x)) for x in tgt):
                        events.append((name + " host->device", tuple(self.shape)))
            return real(self, *a, **k)

        monkeypatch.setattr(torch.Tensor, name, f)

    for nm in ("cpu", "item", "tolist", "numpy"):
        watch(nm, "d2h")
    for nm in ("to", "cuda"):
        watch(nm, "to")

    sal = st._compute_sal_dict(loader, lambda images: next(it))
    assert inside["bodies"] == nb and inside["depth"] == 0
    assert events == [], events
    assert len(sal["al_metric"]) == nb * b and list(sal["al_metric"])[0] == "7-0"
