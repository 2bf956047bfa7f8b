"""bench.py's launch contract, checked without a GPU: `--gpus N` on a node with fewer than N devices must fail loudly
(not report n_gpus = 1), and under torchrun a WORLD_SIZE that disagrees with --gpus is an error too."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_more_gpus_than_devices_is_an_error():
    r = _run(["--gpus", "2", "--steps", "1"])
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr and '"n_gpus"' not in r.stdout


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and '"n_gpus"' not in r.stdout


def test_pool_flag_is_for_pool_workloads():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert {"c1", "c2", "c3", "c4", "c5"} <= set(m.WORKLOADS)
    assert m.WORKLOADS["c5"]["picks"] == 100 and m.WORKLOADS["c4"]["score"] == "MPE"
    assert m.SPLIT_PRODUCTS == {"p2": 3, "h2": 3, "bf3": 6}


def _bench_mod():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_fits_the_driver_tail():
    """The line the driver parses: built by the same function main() prints through, from full records of every workload
    (round 5's 17-22 KB lines are the canned results), must be ONE line of at most 4 096 characters, strict JSON, and
    carry the contract's fields, the roofline and the CPU baseline."""
    import glob
    import json

    m = _bench_mod()
    recs = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "bench_records", "*.json")))
    assert len(recs) >= 4
    for p in recs:
        with open(p) as f:
            full = json.load(f)
        line = m.compact_line(full, "bench_detail.json")
        assert "\n" not in line and len(line) <= 4096, (p, len(line))
        d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # strict: no NaN / Infinity
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in d, (p, k)
        assert d["value"] == round(full["value"], 4) and d["ms_per_step"] == round(full["ms_per_step"], 4)
        assert len(d["dtype"]) <= 120 and "workload" in d["config"]
        r = d["roofline"]
        if r is not None:
            assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and len(r["kernel"]) <= 60
            if r["frac"] is not None:
                assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
        if "cpu_baseline" in full:
            cb = d["cpu_baseline"]
            assert set(("value", "unit", "cores", "kind", "sample", "parity_sample")) <= set(cb)
            assert all(isinstance(x, (int, float, bool)) for x in cb["parity_sample"].values())
        if "companions" in full:
            for name in ("c3", "c4", "c2_with_input"):
                assert set(("ms_per_step", "value", "frac")) <= set(d["companions"][name])
        if "exact_modes" in full:
            assert set(d["exact_modes"]) == {"h2", "bf3", "fp32"}


def test_compact_line_survives_non_finite_and_oversized_input():
    import json

    m = _bench_mod()
    full = {"metric": "m", "value": float("nan"), "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": float("inf"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 " + "x" * 500, "data": "synthetic",
            "config": {"workload": "w" * 1000, "parallelism": "p" * 400},
            "roofline": {"kernel": "k" * 900, "bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.25e-4, "traffic": None,
                         "other_kernels": [{"kernel": "z" * 5000}] * 40},
            "companions": {n: {"error": "e" * 2000} for n in ("c3", "c4", "c2_with_input")},
            "attribution": {"per_rank_s": {"min": 1, "mean": 1, "max": 1, "all": [1.0] * 64}, "compute_s": {"min": 1, "max": 2}}}
    line = m.compact_line(full, "d.json")
    assert len(line) <= 4096
    d = json.loads(line)
    assert d["value"] is None and d["ms_per_step"] is None and len(d["roofline"]["kernel"]) <= 60 and "all" not in d["attribution"]["per_rank_s"]
