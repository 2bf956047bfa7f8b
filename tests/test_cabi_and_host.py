"""CPU (no GPU): the C-ABI library builds/loads and exports every symbol that
include/mval_hip.h declares; host logic (graph, plan, config, selection) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import cases

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from multi_view_active_learning_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(REPO, "include", "mval_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mval_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/mval_hip.h but not exported"
    assert lib.mval_version() >= 100
    assert lib.mval_packed_weight_floats(ctypes.c_int(1), ctypes.c_int(19), ctypes.c_int(32), ctypes.c_int(1)) == 2 * 2 * 256
    assert lib.mval_packed_weight_floats(ctypes.c_int(0), ctypes.c_int(64), ctypes.c_int(3), ctypes.c_int(7)) == 49 * 3 * 64


def test_product_path_has_no_cpu_fallback():
    from multi_view_active_learning_amd import _lib
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet
    from multi_view_active_learning_amd.utils.triangulation import triangulation

    with pytest.raises(_lib.MvalError):
        triangulation(torch.zeros(4, 19, 8, 8), torch.zeros(4, 3, 4), 4, torch.ones(19))
    with pytest.raises(_lib.MvalError):
        PoseHighResolutionNet(19).eval()(torch.zeros(1, 3, 64, 64))
    # the product package never imports the oracle
    import subprocess, sys
    code = "import sys; import multi_view_active_learning_amd.strategy, multi_view_active_learning_amd.engine, multi_view_active_learning_amd.ops; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle leaked into the product'"
    subprocess.check_call([sys.executable, "-c", code], cwd=REPO)


def test_graph_counts_match_survey():
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet

    g = PoseHighResolutionNet(19)._graph
    assert sum(o.kind == "conv" for o in g.ops) == 293
    assert sum(o.bn is not None for o in g.ops) == 292
    assert sum(o.up > 0 for o in g.ops) == 28
    r = PoseResNet(19)._graph
    assert sum(o.kind == "conv" for o in r.ops) == 54 and sum(o.kind == "deconv" for o in r.ops) == 3
    assert len(PoseHighResolutionNet(19).state_dict()) == 1754 and len(PoseResNet(19).state_dict()) == 338
    with pytest.raises(NotImplementedError):
        PoseResNet(19, 18)


@pytest.mark.parametrize("name", ["w32", "w48", "r50"])
def test_plan_geometry_and_arena(lib, name):
    from multi_view_active_learning_amd import engine

    c = cases.model_cases()[name]
    m = cases.product_model(c)
    p = engine.InferencePlan(m, 4, c["h"], c["w"], torch.device("cpu"))
    assert p.out_hw == (c["h"] // 4, c["w"] // 4) and p.out_channels == c["j"]
    # no op may read and write overlapping arena ranges
    g = p.graph  # (the plan's own op order: a P2 plan regroups the fuse layers' first-level stride-2 convs)
    for i, op in enumerate(g.ops):
        o = p.graph_ops[i]
        if o.out_off < 0:
            continue
        out_n = 4 * (o.hout << o.up) * (o.wout << o.up) * o.cout
        for off, a in ((o.in_off, op.src), (o.res1_off, op.res1), (o.res2_off, op.res2)):
            if off < 0 or a is None:
                continue
            act = g.acts[a]
            n_in = 4 * (c["h"] // act.down) * (c["w"] // act.down) * act.channels
            assert off + n_in <= o.out_off or o.out_off + out_n <= off, (i, op.conv)
        assert o.out_off + out_n <= p.arena_floats
    with pytest.raises(ValueError):
        engine.InferencePlan(cases.product_model(cases.model_cases()["w32"]), 1, 100, 100, torch.device("cpu"))


def test_p2_launch_list_structure(lib, monkeypatch):
    """The launch list of a P2 plan (built on the host, no kernel runs): HRNet-W32 at 256 x 256 becomes fused stem + downsample conv
    + 4 fused Bottlenecks + 32 fused BasicBlocks + 9 fused up-path launches + single convs; every fused op points at the graph's
    tensors (same arena offsets as the ops it replaces); each switch restores the op-by-op form; W48 / PoseResNet keep the h2 plan."""
    from multi_view_active_learning_amd import engine

    monkeypatch.setenv("MVAL_CONV", "p2")
    c = cases.model_cases()["w32"]
    m = cases.product_model(c)
    g = m._graph

    def plan():
        return engine.InferencePlan(m, 4, 256, 256, torch.device("cpu"))

    p = plan()
    assert p.p2
    kinds = [o.kind for o in p.ops]
    assert kinds[0] == engine.OP_STEM_P2 and kinds[1] == engine.OP_CONV and kinds[2] == engine.OP_BNECK
    assert (kinds.count(engine.OP_BNECK), kinds.count(engine.OP_BLOCK), kinds.count(engine.OP_FUSE_UP), kinds.count(engine.OP_TO_P2)) == (4, 32, 9, 0)
    # launches = graph ops - (stem pair: 1) - (bottlenecks: 2 each) - (blocks: 1 each) - (up-path chains: terms - 1 each)
    fu = [o for o in p.ops if o.kind == engine.OP_FUSE_UP]
    assert len(p.ops) == len(g.ops) - 1 - 8 - 32 - sum(o.n_terms - 1 for o in fu)
    # every fused op has rows for its output; an up-path launch holds consecutive terms (factors 2, 4[, 8] for branch 0; 2, 4 for
    # branch 1), each from the branch with cout << up channels, and carries the fuse output's ReLU
    for o in p.ops:
        if o.kind in (engine.OP_BNECK, engine.OP_BLOCK, engine.OP_FUSE_UP, engine.OP_STEM_P2):
            assert o.out_off >= 0 and o.out_amax_off > 0
        if o.kind == engine.OP_FUSE_UP:
            ups = [o.t_up[j] for j in range(o.n_terms)]
            assert ups == list(range(1, 1 + o.n_terms)) and o.relu == 1 and o.res1_off >= 0 and o.res1_amax_off > 0
            assert [o.t_cin[j] for j in range(o.n_terms)] == [o.cout << u for u in ups]
            assert all(o.t_in_off[j] >= 0 and o.t_in_amax_off[j] > 0 and o.t_bound_off[j] > 0 for j in range(o.n_terms))
    assert sorted((o.cout, o.n_terms) for o in fu) == sorted([(32, 2)] * 4 + [(32, 3)] * 3 + [(64, 2)] * 2)
    for var, kind in (("MVAL_P2_STEM", engine.OP_STEM_P2), ("MVAL_P2_BNECK", engine.OP_BNECK), ("MVAL_P2_FUSE_UP", engine.OP_FUSE_UP)):
        monkeypatch.setenv(var, "0")
        assert kind not in [o.kind for o in plan().ops]
        monkeypatch.delenv(var)
    monkeypatch.setenv("MVAL_FUSE_BLOCKS", "0")
    assert {o.kind for o in plan().ops} == {engine.OP_CONV, engine.OP_TO_P2}
    monkeypatch.delenv("MVAL_FUSE_BLOCKS")
    # PoseResNet (max-pool, transposed convs) never; HRNet-W48 since round 4 (odd full-width tiles, 48- / 96-channel fused up-paths),
    # MVAL_P2_W48=0 keeps it on the h2 kernels
    cc = cases.model_cases()["r50"]
    assert not engine.InferencePlan(cases.product_model(cc), 2, cc["h"], cc["w"], torch.device("cpu")).p2
    cc = cases.model_cases()["w48"]
    p48 = engine.InferencePlan(cases.product_model(cc), 2, cc["h"], cc["w"], torch.device("cpu"))
    assert p48.p2 and sorted((o.cout, o.n_terms) for o in p48.ops if o.kind == engine.OP_FUSE_UP) == sorted([(48, 2)] * 4 + [(48, 3)] * 3 + [(96, 2)] * 2)
    assert sum(o.kind == engine.OP_BNECK for o in p48.ops) == 4 and p48.ops[0].kind == engine.OP_STEM_P2
    monkeypatch.setenv("MVAL_P2_W48", "0")
    assert not engine.InferencePlan(cases.product_model(cc), 2, cc["h"], cc["w"], torch.device("cpu")).p2


def test_config_tree_and_factory():
    from multi_view_active_learning_amd.config import get_default_configs
    from multi_view_active_learning_amd.pose_estimators import get_pose_net, PoseResNet, PoseHighResolutionNet

    cfg = get_default_configs()
    assert cfg.AL.ITER_AMOUNT == 100 and cfg.TRAIN.LOSS_CLIP_VALUE == 10.0 and cfg.POSE_ESTIMATOR.STRIDE == 4
    assert isinstance(get_pose_net(cfg), PoseResNet)
    cfg.merge_from_list(["POSE_ESTIMATOR.TYPE", "HRNET", "DATA.NUM_JOINTS", 42])
    net = get_pose_net(cfg)
    assert isinstance(net, PoseHighResolutionNet) and net.num_joints == 42
    c2 = cfg.clone()
    c2.AL.STRATEGY = "HP"
    assert cfg.AL.STRATEGY == "RANDOM"
    with pytest.raises(KeyError):
        cfg.merge_from_list(["AL.NOPE", 1])


def test_tables_to_sal_dict_gather_order():
    from multi_view_active_learning_amd.strategy import tables_to_sal_dict

    j = 2

    def tab(rows):
        return np.asarray([[p, f, a, s, i, m] + [0.5] * (3 * j) for (p, f, a, s, i, m) in rows], dtype=np.float64)

    r0 = tab([(0, 0, 1.0, 2.0, 3, 4.0), (0, 1, 1.5, 2.0, 3, 4.0), (0, 2, 1.7, 2.0, 4, 4.0)])
    r1 = tab([(1, 0, 9.0, 2.0, 3, 4.0), (1, 1, 8.0, 2.0, 2, float("nan")), (1, 2, 7.0, 2.0, 3, 4.0)])
    d = tables_to_sal_dict([r0, r1], [2, 1])
    # batch 0: sample 0 (rank0, rank1), sample 1 (rank0, rank1); batch 1: sample 0 (rank0, rank1)
    assert list(d["al_metric"]) == ["0-0", "1-0", "0-1", "1-1", "0-2", "1-2"]
    assert d["inlier_count"]["1-1"] == 2.0 and np.isnan(d["mkpe"]["1-1"])
    assert d["pred_3d_keypoints"]["0-0"] == [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]


# ---- on-disk formats: tests/golden/formats.json holds what the reference's own writers produced ----
def _formats():
    import json

    with open(os.path.join(os.path.dirname(__file__), "golden", "formats.json")) as f:
        return json.load(f)


def test_iteration_files_match_reference_text(tmp_path):
    """write_iteration leaves the same files, byte for byte, as the reference's rank-0 branch of sample_next_batch
    (strategy.py:54-135) did for the same guids / score dictionaries."""
    import json

    from multi_view_active_learning_amd.utils import experiment_io as eio

    g = _formats()
    sal = json.loads(g["files"]["SAL-DICT-ITER-1"])
    eio.write_iteration(str(tmp_path), "expr", 0, g["seed_guids"])
    eio.write_iteration(str(tmp_path), "expr", 1, g["al_guids"], g["sal_guids"], sal)
    got = {n: open(os.path.join(tmp_path, "expr", n)).read() for n in sorted(os.listdir(tmp_path / "expr"))}
    assert got == g["files"]
    # an empty pseudo-label list writes no SAL-GUID file (strategy.py:73: `if len(sal_guids) != 0`)
    eio.write_iteration(str(tmp_path), "expr", 2, g["al_guids"], [], sal)
    assert not os.path.exists(eio.sal_guid_path(str(tmp_path), "expr", 2))
    assert os.path.exists(eio.sal_dict_path(str(tmp_path), "expr", 2))


def test_sal_dict_text_from_packed_tables():
    """The packed per-rank table -> five dicts -> json.dumps reproduces the reference's SAL-DICT text exactly: the
    values leave the table as python floats of the same fp32 / fp64 numbers the reference's ``.data.item()`` gives."""
    import json

    from multi_view_active_learning_amd.strategy import tables_to_sal_dict

    g = _formats()
    text = g["files"]["SAL-DICT-ITER-1"]
    sal = json.loads(text)
    guids = list(sal["al_metric"])
    rows = []
    for guid in guids:
        pose, frame = guid.split("-")
        rows.append([float(pose), float(frame), sal["al_metric"][guid], sal["sal_metric"][guid], sal["inlier_count"][guid],
                     sal["mkpe"][guid]] + list(np.asarray(sal["pred_3d_keypoints"][guid]).reshape(-1)))
    table = np.asarray(rows, dtype=np.float64)
    assert json.dumps(tables_to_sal_dict([table], [2, 2])) == text


def test_restore_guids_matches_reference_restore_dataset(tmp_path):
    from multi_view_active_learning_amd.utils import experiment_io as eio

    g = _formats()
    for name, text in g["files"].items():
        p = tmp_path / "expr" / name
        p.parent.mkdir(exist_ok=True)
        p.write_text(text)
    labeled, pseudo = eio.restore_guids(str(tmp_path), "expr", 2, expr_type="SAL")
    assert labeled == g["restore"]["labeled"] and pseudo == g["restore"]["pseudo"]
    assert eio.restore_guids(str(tmp_path), "expr", 2, expr_type="AL")[1] is None
    assert eio.restore_guids(str(tmp_path), "expr", 1, expr_type="SAL") == ([g["seed_guids"]], None)
    assert eio.read_sal_dict(eio.sal_dict_path(str(tmp_path), "expr", 1))["al_metric"] == __import__("json").loads(
        g["files"]["SAL-DICT-ITER-1"])["al_metric"]


@pytest.mark.parametrize("kind", ["POSE_RESNET", "HRNET"])
def test_checkpoint_structure_matches_reference(tmp_path, kind):
    """save_checkpoint writes what _save_checkpoints (strategy.py:681-711) writes: same file name, top-level keys,
    state_dict names / shapes / dtypes IN THE SAME ORDER (the optimizer state is indexed by that order), same Adam
    state layout; and it loads back strictly, with or without the DistributedDataParallel ``module.`` prefix."""
    import hashlib

    import torch

    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet
    from multi_view_active_learning_amd.utils import experiment_io as eio

    g = _formats()["checkpoint"][kind]
    assert g["reference_loads_ours"] and g["ours_loads_reference"]  # checked live by make_golden.py
    make = (lambda: PoseResNet(19, 50)) if kind == "POSE_RESNET" else (lambda: PoseHighResolutionNet(19))
    model = make()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    path = eio.save_checkpoint(str(tmp_path / "checkpoints"), 3, 17, model, opt)
    assert os.path.basename(path) == g["file"] == eio.checkpoint_name(3, 17)
    assert eio.checkpoint_name(3, 17, mkpe=41.256) == "CKPT-E17-MKPE41.26.pth"
    blob = eio.load_checkpoint(path)
    assert list(blob.keys()) == g["top_keys"] and (blob["epoch"], blob["global_step"]) == (g["epoch"], g["global_step"])
    lines = "".join("%s:%s:%s\n" % (k, list(v.shape), v.dtype) for k, v in blob["state_dict"].items())
    assert len(blob["state_dict"]) == g["n_entries"]
    assert hashlib.sha256(lines.encode()).hexdigest() == g["state_dict_sha256"]
    if "state_dict" in g:
        assert [[k, list(v.shape), str(v.dtype)] for k, v in blob["state_dict"].items()] == g["state_dict"]
    osd = blob["optimizer"]
    assert sorted(osd["param_groups"][0].keys()) == g["param_group_keys"]
    assert sorted(osd["state"][0].keys()) == g["optimizer_state_keys"]
    assert len(osd["param_groups"][0]["params"]) == g["n_optimizer_params"]
    # overwrite in place, then restore into a fresh model + optimizer
    assert eio.save_checkpoint(str(tmp_path / "checkpoints"), 3, 17, model, opt) == path
    fresh = make()
    fopt = torch.optim.Adam(fresh.parameters(), lr=1e-3)
    assert eio.restore_checkpoint(path, fresh, fopt) == (3, 17)
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))
    assert all(fopt.state[p]["exp_avg"].shape == p.shape for p in fresh.parameters())
    # a checkpoint written from a DDP wrapper ("module." keys) loads into a bare model and the other way round
    wrapped = {"module." + k: v for k, v in blob["state_dict"].items()}
    torch.save(dict(blob, state_dict=wrapped), str(tmp_path / "ddp.pth"))
    assert eio.load_weights(make(), restore_from=str(tmp_path / "ddp.pth")) == "restored"

    class Wrapper(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m

    assert eio.load_weights(Wrapper(make()), restore_from=path) == "restored"
    assert eio.load_weights(make()) == "scratch"


def test_init_weight_filtering(tmp_path):
    """_load_weights' INIT_WEIGHT branch (strategy.py:723-741): PoseResNet drops the final layer of the pretrained
    file; HRNet keeps only entries under ``pretrained_layers``."""
    import torch

    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet
    from multi_view_active_learning_amd.utils import experiment_io as eio

    src = PoseResNet(19, 50)
    with torch.no_grad():
        for p in src.parameters():
            p.fill_(0.25)
    torch.save(src.state_dict(), str(tmp_path / "r50.pth"))
    dst = PoseResNet(19, 50)
    before = dst.state_dict()["final_layer.weight"].clone()
    assert eio.load_weights(dst, init_weight=str(tmp_path / "r50.pth"), estimator_type="POSE_RESNET") == "initialized"
    sd = dst.state_dict()
    assert torch.equal(sd["final_layer.weight"], before) and float(sd["conv1.weight"].flatten()[0]) == 0.25

    hsrc = PoseHighResolutionNet(19)
    with torch.no_grad():
        for p in hsrc.parameters():
            p.fill_(0.5)
    torch.save(hsrc.state_dict(), str(tmp_path / "hr.pth"))
    hdst = PoseHighResolutionNet(19)
    layers = hdst.pretrained_layers
    keep_all = layers[0] == "*"
    eio.load_weights(hdst, init_weight=str(tmp_path / "hr.pth"), estimator_type="HRNET")
    for k, v in hdst.state_dict().items():
        if k.endswith("weight") and v.dim() == 4:
            loaded = float(v.flatten()[0]) == 0.5
            assert loaded == (keep_all or k.split(".")[0] in layers), k
    with pytest.raises(ValueError):
        eio.load_weights(hdst, init_weight=str(tmp_path / "hr.pth"), estimator_type="OTHER")


def test_cluster_file_features(tmp_path):
    import json

    from multi_view_active_learning_amd.utils import experiment_io as eio

    rng = np.random.default_rng(5)
    poses = {"3-%d" % i: rng.standard_normal((4, 19)).tolist() for i in range(6)}
    (tmp_path / "clusters.json").write_text(json.dumps(poses))
    f = eio.read_cluster_features(str(tmp_path / "clusters.json"), 2)
    assert f.shape == (6, 57)
    kp = np.array(poses["3-4"])
    assert np.array_equal(f[4], (kp[:3] - kp[:3, 2:3]).flatten()) and np.all(f[:, 2] == 0)


def test_al_dict_for_coreset_layout():
    """dataset/dataset.py:47-51: records hold (>=3, J) poses, the core-set wants (J, >=3) rows keyed by index."""
    from multi_view_active_learning_amd.utils.coreset import get_al_dict_for_coreset

    rng = np.random.default_rng(2)
    labeled = [{"3d_keypoints": rng.standard_normal((4, 19)).tolist()} for _ in range(3)]
    d = get_al_dict_for_coreset(labeled)
    assert list(d) == [0, 1, 2] and d[1].shape == (19, 4) and d[1].dtype == np.float64
    assert np.array_equal(d[2], np.array(labeled[2]["3d_keypoints"]).T)


def test_argmax_key_stash_follows_the_network_output():
    """Decode from the heat-map layer's epilogue: the keys a forward kept are found for the tensor it returned
    and for full views of it, not for slices, copies or tensors written to since, and die with the tensor."""
    import gc

    import torch

    from multi_view_active_learning_amd import _lib

    out = torch.zeros(8, 17, 4, 4)
    keys = torch.zeros((8 * 17, _lib.ARGMAX_SLOTS), dtype=torch.int64)
    _lib.remember_argmax_keys(out, keys)
    assert _lib.argmax_keys_of(out) is keys
    assert _lib.argmax_keys_of(out.reshape(2, 4, 17, 4, 4)) is keys
    assert _lib.argmax_keys_of(out.reshape(2, 4, 17, 4, 4).to(torch.float32).contiguous()) is keys
    assert _lib.argmax_keys_of(out[:4]) is None and _lib.argmax_keys_of(out[4:]) is None
    assert _lib.argmax_keys_of(out.clone()) is None
    assert _lib.argmax_keys_of(out.permute(0, 1, 3, 2)) is None
    view = out.reshape(2, 4, 17, 4, 4)
    view[0, 0, 0, 0, 0] = 1.0  # written through a view: the keys no longer describe the tensor
    assert _lib.argmax_keys_of(out) is None and _lib.argmax_keys_of(view) is None
    with torch.inference_mode():  # no version counter to trust: decoded from the maps
        inf = torch.zeros(8, 17, 4, 4)
        _lib.remember_argmax_keys(inf, keys)
        assert _lib.argmax_keys_of(inf) is None
    del inf
    n = len(_lib._ARGMAX_KEYS)
    del out, view
    gc.collect()
    assert len(_lib._ARGMAX_KEYS) == n - 1


def test_argmax_keys_serve_only_the_plans_factorisation(monkeypatch):
    """The keys are [image][slot][joint of the plan]: a caller that factors the same heat-maps differently (b = n * J, v = 1,
    j = 1 has the same element count) must be decoded from the maps, not from mis-indexed keys; and
    forget_argmax_keys drops them for callers that write into the output through its data pointer."""
    import torch

    from multi_view_active_learning_amd import _lib

    used = []
    monkeypatch.setattr(_lib, "argmax_from_keys", lambda *a, **k: used.append("keys") or "from-keys")

    class _FakeLib:
        def mval_argmax_decode(self, *a):
            used.append("maps")
            return 0

    monkeypatch.setattr(_lib, "lib", lambda: _FakeLib())
    monkeypatch.setattr(_lib, "_p", lambda t: None)
    monkeypatch.setattr(_lib, "_stream", lambda: None)
    n, J, h, w = 8, 17, 4, 4
    out = torch.zeros(n, J, h, w)
    keys = torch.zeros((n, _lib.ARGMAX_SLOTS, J), dtype=torch.int64)
    _lib.remember_argmax_keys(out, keys)
    valid = torch.ones(2, J, dtype=torch.uint8)
    assert _lib.argmax_decode(out.reshape(2, 4, J, h, w), valid, 2, 4, J, h, w, 4, h) == "from-keys"
    _lib.argmax_decode(out.reshape(n * J, 1, 1, h, w), torch.ones(n * J, 1, dtype=torch.uint8), n * J, 1, 1, h, w, 4, h)
    _lib.argmax_decode(out.reshape(1, n, J, h, w), valid, 1, n, J, h, w, 4, h)  # (b * v = n, j = J: same layout -> keys)
    assert used == ["keys", "maps", "keys"]
    _lib.forget_argmax_keys(out.reshape(2, 4, J, h, w))
    assert _lib.argmax_keys_of(out) is None


def test_python_constants_match_the_header():
    """The ctypes mirror restates a few #defines / enum values of include/mval_hip.h: they must agree."""
    import os
    import re

    from multi_view_active_learning_amd import _lib, engine

    text = open(os.path.join(os.path.dirname(__file__), "..", "include", "mval_hip.h")).read()

    def define(name):
        return int(re.search(r"#define\s+%s\s+(\d+)" % name, text).group(1))

    def enum(name):
        return int(re.search(r"\b%s\s*=\s*(\d+)" % name, text).group(1))

    assert define("MVAL_ARGMAX_SLOTS") == _lib.ARGMAX_SLOTS
    assert define("MVAL_AMAX_ROW") == engine.AMAX_ROW
    from multi_view_active_learning_amd import engine_train

    assert define("MVAL_TRAIN_LANE_FWD") == engine_train.TRAIN_LANE_FWD and define("MVAL_TRAIN_LANE_BWD") == engine_train.TRAIN_LANE_BWD
    assert define("MVAL_TRAIN_LANE_ORD") == engine_train.TRAIN_LANE_ORD and define("MVAL_TRAIN_BSUM") == engine_train.TRAIN_BSUM and define("MVAL_TRAIN_WGRAD_DEFER") == engine_train.TRAIN_WGRAD_DEFER
    # the ctypes mirror of mval_train_op ends with the round-6 fields and has the header's size (8-byte aligned, two int32 at the end)
    assert [f[0] for f in engine_train.MvalTrainOp._fields_][-2:] == ["zin_rel", "z_out"]
    csrc = open(os.path.join(os.path.dirname(__file__), "..", "multi_view_active_learning_amd", "csrc", "conv_common.h")).read()
    assert int(re.search(r"#define\s+MVAL_MAX_LANES\s+(\d+)", csrc).group(1)) == engine_train.MAX_LANES
    for name, val in (("MVAL_OP_BNECK", engine.OP_BNECK), ("MVAL_OP_STEM_P2", engine.OP_STEM_P2), ("MVAL_OP_FUSE_UP", engine.OP_FUSE_UP),
                      ("MVAL_OP_BLOCK", engine.OP_BLOCK)):
        assert enum(name) == val, name


def test_training_lane_flags_keep_every_gradient_slot_on_one_ordered_chain():
    """engine_train.lane_flags (host logic behind mval_train_*_lanes, hrnet.py:199-287): on HRNet-W32 / -W48 graphs, in every mode --
    an op runs its backward on a side lane UNORDERED only in phases where each gradient slot it writes has all its writers on that lane;
    in a phase whose lanes share a slot EVERY op (lane 0 included) carries the ordering bit (modes 2, 3) or the side lanes stay on the
    caller's stream (mode 1); the join-less mode marks every op; PoseResNet (one lane) gets no lane bit."""
    from multi_view_active_learning_amd import engine_train as et
    from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet, hrnet_w48

    for model in (PoseHighResolutionNet(19), PoseHighResolutionNet(5, hrnet_cfg=hrnet_w48())):
        g = model._graph
        nl = min(et.MAX_LANES, max(op.lane for op in g.ops) + 1)
        assert nl == 4
        for mode in ("1", "2", "3"):
            fl = et.lane_flags(g, nl, mode)
            assert len(fl) == len(g.ops)
            by_phase = {}
            for op, f in zip(g.ops, fl):
                by_phase.setdefault(op.phase, []).append((op, f))
            n_free_lane, n_ord = 0, 0
            for ph, ops in by_phase.items():
                slot_lanes = {}
                for op, f in ops:
                    lane = op.lane if (f & et.TRAIN_LANE_BWD) else 0  # the stream its backward runs on
                    for a in (None if op.src == g.input else op.src, op.res1, op.res2):
                        if a is not None:
                            slot_lanes.setdefault(a, set()).add(lane)
                shared = any(len(v) > 1 for v in slot_lanes.values())
                for op, f in ops:
                    assert bool(f & et.TRAIN_LANE_FWD) == (0 < op.lane < nl)
                    assert bool(f & et.TRAIN_LANE_FREE) == (mode == "3")
                    if shared:  # slots written from several streams: every writer of the phase must be ordered
                        assert f & et.TRAIN_LANE_ORD, (mode, ph)
                    n_ord += bool(f & et.TRAIN_LANE_ORD)
                    n_free_lane += bool((f & et.TRAIN_LANE_BWD) and not (f & et.TRAIN_LANE_ORD))
                if mode == "1":
                    assert not shared  # (shared phases keep their side lanes on the caller's stream)
            assert n_free_lane > 100 and (n_ord > 40) == (mode != "1")
    g = PoseResNet(19)._graph
    assert max(op.lane for op in g.ops) == 0 and not any(et.lane_flags(g, 1, "3")[i] & (et.TRAIN_LANE_FWD | et.TRAIN_LANE_BWD) for i in range(len(g.ops)))


def test_adam_mirror_is_torch_adam_off_the_device():
    """multi_view_active_learning_amd.optim.Adam (strategy.py:405-407, :479): on CPU tensors -- and for every configuration the kernel
    does not implement -- it IS torch.optim.Adam: same updates bit for bit, same state_dict layout, StepLR drives it."""
    from multi_view_active_learning_amd.optim import Adam, _JOB

    assert _JOB.itemsize == 40  # = sizeof(mval_adam_job)
    torch.manual_seed(0)
    ws = [torch.randn(5, 3), torch.randn(7), torch.randn(2, 2, 3, 3)]
    a = [torch.nn.Parameter(w.clone()) for w in ws]
    b = [torch.nn.Parameter(w.clone()) for w in ws]
    oa = Adam([{"params": a, "lr": 1e-2}], weight_decay=0.01)
    ob = torch.optim.Adam([{"params": b, "lr": 1e-2}], weight_decay=0.01)
    sa = torch.optim.lr_scheduler.StepLR(oa, step_size=2)
    sb = torch.optim.lr_scheduler.StepLR(ob, step_size=2)
    for it in range(5):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
    da, db = oa.state_dict(), ob.state_dict()
    assert da["param_groups"][0]["lr"] == db["param_groups"][0]["lr"]
    assert da["state"].keys() == db["state"].keys()
    for k in da["state"]:
        assert da["state"][k].keys() == db["state"][k].keys()
        for n in da["state"][k]:
            assert torch.equal(da["state"][k][n], db["state"][k][n])


def test_param_signature_sees_registrations_after_the_first_walk():
    """engine._param_signature caches (dict, key) slots per model; a parameter or buffer registered later, a bias set from None and a
    swapped submodule must still be seen -- the slots are rebuilt when nn.Module's registration hooks have fired."""
    import copy

    import torch

    from multi_view_active_learning_amd import engine
    from multi_view_active_learning_amd.pose_estimators import PoseResNet

    m = PoseResNet(19, 50)
    s0 = engine._param_signature(m)
    assert engine._param_signature(m) == s0
    holder = next(h for h in m._holders.values() if hasattr(h, "bias") and h.bias is None)  # a conv without bias
    holder.bias = torch.nn.Parameter(torch.zeros(holder.weight.shape[0]))  # bias set from None
    s1 = engine._param_signature(m)
    assert len(s1) == len(s0) + 2  # one more version counter, one more data pointer
    holder.register_buffer("extra_stat", torch.zeros(3))
    s2 = engine._param_signature(m)
    assert len(s2) == len(s1) + 1
    del holder._buffers["extra_stat"]  # (no hook fires on deletion: the KeyError path rebuilds)
    assert len(engine._param_signature(m)) == len(s1)
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        next(m2.parameters()).add_(1.0)
    assert engine._param_signature(m2) != engine._param_signature(m)  # the copy's slots are its own
    assert engine._param_signature(m) == s1
