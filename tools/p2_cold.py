#!/usr/bin/env python3
"""Launch time of one P2 conv shape with its operands warm in the 256 MB infinity cache (the same tensors every launch, what
tools/p2_sweep.py times) and cold (a rotation of operand sets larger than the cache, what the kernel sees inside the network).
usage: p2_cold.py cin cout h w k stride [n=128] [sets=8]     (with a -DP2_STAMP build: phase table of one cold launch)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib, ops

cin, cout, h, w, k, stride = (int(v) for v in sys.argv[1:7])
n = int(sys.argv[7]) if len(sys.argv) > 7 else 128
sets = int(sys.argv[8]) if len(sys.argv) > 8 else 8
dev = torch.device("cuda:0")
lib = _lib.lib()
ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
convs = []
for _ in range(sets):
    x = torch.relu(torch.randn(n, h, w, cin, device=dev))
    r = torch.randn(n, ho, wo, cout, device=dev)
    convs.append(ops.P2Conv(x, wt, one, zero, stride=stride, relu=True, res1=r))
mb = n * (h * w * cin + 2 * ho * wo * cout) * 4 / 1e6


def loop(fns, reps=200):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


warm = loop([convs[0].launch])
cold = loop([c.launch for c in convs])
fl = 2.0 * n * ho * wo * cin * cout * k * k
print(f"{cin}->{cout} {h}x{w} k{k} s{stride} n={n}: {mb:.0f} MB per launch; warm {warm:.1f} us ({fl / warm / 1e6:.0f} TFLOP/s, {mb / warm / 1e3:.2f} TB/s)"
      f"  cold ({sets} sets) {cold:.1f} us ({fl / cold / 1e6:.0f} TFLOP/s, {mb / cold / 1e3:.2f} TB/s)")
if hasattr(lib, "mval_p2_debug_buffer") and os.environ.get("P2_STAMPS"):
    dbg = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
    for c in convs:
        c.launch()
    torch.cuda.synchronize()
    lib.mval_p2_debug_buffer(C.c_void_p(dbg.data_ptr()))
    convs[0].launch()
    torch.cuda.synchronize()
    lib.mval_p2_debug_buffer(C.c_void_p(0))
    d = dbg.cpu().numpy().reshape(-1, 16)
    d = d[d[:, 0] != 0]
    t0 = d[:, 0].min()
    print(f"  cold launch: {len(d)} waves; span {(d[d > 0].max() - t0) / 100:.1f} us; first barrier at median +{np.median((d[:, 1] - d[:, 0]) / 100.0):.2f} us")
    names = ["mfma (all but last stage)", "store + barrier (inner)", "next tile plan + loads", "prefetch + mfma (last stage)", "epilogue", "store + barrier (tile end)"]
    life = (d[:, 4] - d[:, 0]) / 100.0
    for kk in range(6):
        v = d[:, 8 + kk] / 100.0
        print(f"    {names[kk]:30s} {np.median(v):7.2f} us per wave = {100 * np.median(v) / np.median(life):5.1f} %")
    print("    wave lifetime median %.2f us" % np.median(life))
