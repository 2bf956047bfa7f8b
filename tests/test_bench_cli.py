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
