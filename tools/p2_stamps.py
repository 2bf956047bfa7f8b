#!/usr/bin/env python3
"""Phase timeline of the P2 conv kernel from in-kernel wall-clock stamps (diagnostic build only:
MVAL_EXTRA_CFLAGS=-DP2_STAMP python -m multi_view_active_learning_amd.build --force).
usage: p2_stamps.py cin cout h w k stride [n=128]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib, ops

cin, cout, h, w, k, stride = (int(v) for v in sys.argv[1:7])
n = int(sys.argv[7]) if len(sys.argv) > 7 else 128
dev = torch.device("cuda:0")
lib = _lib.lib()
ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
x = torch.relu(torch.randn(n, h, w, cin, device=dev))
wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
r = torch.randn(n, ho, wo, cout, device=dev)
c = ops.P2Conv(x, wt, one, zero, stride=stride, relu=True, res1=r)
dbg = torch.zeros(1 << 22, dtype=torch.int64, device=dev)
lib.mval_p2_debug_buffer(C.c_void_p(dbg.data_ptr()))
for _ in range(3):
    c.launch()
torch.cuda.synchronize()
dbg.zero_()
c.launch()
torch.cuda.synchronize()
lib.mval_p2_debug_buffer(C.c_void_p(0))
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
print(f"{cin}->{cout} {h}x{w} k{k} s{stride} n={n}: {len(d)} waves stamped; kernel span {(d[d > 0].max() - t0) / 100:.1f} us")
rel = (d - t0) / 100.0
print("start of wave (us):  median %.2f  max %.2f ; first barrier passed at median +%.2f us" % (np.median(rel[:, 0]), rel[:, 0].max(), np.median((d[:, 1] - d[:, 0]) / 100.0)))
names = ["mfma (all but last stage)", "store + barrier (inner)", "next tile plan + loads", "prefetch + mfma (last stage)", "epilogue", "store + barrier (tile end)"]
life = (d[:, 4] - d[:, 0]) / 100.0
for kk in range(6):
    v = d[:, 8 + kk] / 100.0
    print(f"  {names[kk]:30s} total {np.median(v):7.2f} us per wave (min {v.min():6.2f} max {v.max():6.2f}) = {100 * np.median(v) / np.median(life):5.1f} % of its life")
print("wave lifetime (us): median %.2f  min %.2f  max %.2f" % (np.median(life), life.min(), life.max()))
