#!/usr/bin/env python3
"""Phase timers of the fused P2 Bottleneck kernel (diagnostic build: MVAL_EXTRA_CFLAGS=-DP2_STAMP).  usage: p2_bneck_stamps.py CIN H W [n=128]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib, ops

cin, h, w = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device("cuda:0")
lib = _lib.lib()
x = torch.relu(torch.randn(n, h, w, cin, device=dev))
convs = []
for co, ci, k in ((64, cin, 1), (64, 64, 3), (256, 64, 1)):
    convs.append((torch.randn(co, ci, k, k, device=dev) * (2.0 / (ci * k * k)) ** 0.5, torch.ones(co, device=dev), torch.zeros(co, device=dev)))
res = None if cin == 256 else torch.randn(n, h, w, 256, device=dev)
b = ops.P2Bneck(x, convs, res)
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
print(f"bottleneck cin{cin} {h}x{w} n={n}: {len(d)} waves; kernel span {(d[:, 4].max() - d[:, 0].min()) / 100:.1f} us; wave lifetime median {np.median(life):.1f} us")
names = ["conv1 MFMAs + chunk staging barriers", "scales, BN1 -> M1", "barrier (M1 complete)", "conv2 MFMAs", "BN2 -> M2", "barrier + conv3 MFMAs (both halves)",
         "BN3 + residual + stores (both halves)", "tile end: barrier, next chunk store"]
for k, nm in enumerate(names):
    v = d[:, 8 + k] / 100.0
    print(f"  {nm:42s} {np.median(v):7.2f} us per wave (min {v.min():6.2f} max {v.max():6.2f}) = {100 * np.median(v) / np.median(life):5.1f} %")
