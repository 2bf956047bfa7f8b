#!/usr/bin/env python3
"""Phase timers of the fused P2 stem kernel (diagnostic build: MVAL_EXTRA_CFLAGS=-DP2_STAMP).  usage: p2_stem_stamps.py [n=128] [h=256] [w=256]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
lib = _lib.lib()
x = torch.randn(n, 3, h, w, device=dev)
w1 = torch.randn(64, 3, 3, 3, device=dev) * (2.0 / 27) ** 0.5
w2 = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
b = ops.P2Stem(x, w1, one, zero, w2, one, zero)
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
lib.mval_p2_debug_buffer(C.c_void_p(dbg.data_ptr()))
for _ in range(3):
    b.launch()
torch.cuda.synchronize()
dbg.zero_()
b.launch()
torch.cuda.synchronize()
lib.mval_p2_debug_buffer(C.c_void_p(0))
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
life = (d[:, 4] - d[:, 0]) / 100.0
print(f"stem {h}x{w} n={n}: {len(d)} waves; kernel span {(d[:, 4].max() - d[:, 0].min()) / 100:.1f} us; wave lifetime median {np.median(life):.1f} us")
names = ["row max, scales, conv1 FMAs", "BN1 -> Y1 (split, LDS writes)", "barrier (Y1 complete)", "patch request + conv2 MFMAs", "BN2 + stores + max", "patch store", "barrier (tile end)"]
for k, nm in enumerate(names):
    v = d[:, 8 + k] / 100.0
    print(f"  {nm:34s} {np.median(v):7.2f} us per wave (min {v.min():6.2f} max {v.max():6.2f}) = {100 * np.median(v) / np.median(life):5.1f} %")
