"""Build csrc/libmval_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m multi_view_active_learning_amd.build [--force] [--save-temps]

One translation unit per .hip file, objects cached under csrc/_obj keyed by mtime, linked
into a single C-ABI shared library that carries no torch dependency.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MVAL_BUILD_TAG=<tag> (measurement): a VARIANT library csrc/libmval_hip_<tag>.so with its own object cache, built with MVAL_EXTRA_CFLAGS
# beside the product library; _lib.py loads it when MVAL_LIB_TAG=<tag> is set (tools/ A/B runs: several variants travel to one GPU call)
TAG = os.environ.get("MVAL_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + TAG if TAG else ""))
LIB = os.path.join(CSRC, "libmval_hip" + ("_" + TAG if TAG else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("MVAL_EXTRA_CFLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header():
    t = os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "mval_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return t


def _compile(src, force, extra):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(sp)
            and os.path.getmtime(obj) > _newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + extra + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    extra = ["-save-temps"] if save_temps else []
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, extra), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[mval build] {LIB} ({os.path.getsize(LIB) // 1024} KiB; {sum(c for _, c in res)} of {len(srcs)} units recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
