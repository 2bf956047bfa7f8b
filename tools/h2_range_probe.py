#!/usr/bin/env python3
"""How the fp16x2-split conv degrades when one value of an image dwarfs the rest (the per-image power-of-two scale puts
the image's max |x| at the top of the fp16 range, so everything else moves down towards the denormals).
For an outlier of 2^k x the typical magnitude: rms error of the outputs the outlier does not touch, against float64,
next to the bf16x3 and exact-fp32 kernels on the same data.  usage: h2_range_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from multi_view_active_learning_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n, c, h, w = 2, 64, 32, 32
x0 = np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32)
wt = (rng.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))).astype(np.float32)
one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
print("outlier   h2 rms      bf3 rms     fp32-mfma rms   (outputs outside the outlier's 3x3 reach, image 0)")
for k in (0, 4, 8, 10, 12, 14, 16, 20, 24):
    x = x0.copy()
    x[0, 5, 3, 3] = 2.0 ** k
    want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, 1, 1)
    mask = torch.ones(h, w, dtype=torch.bool)
    mask[2:5, 2:5] = False
    row = []
    for algo in (ops.ALGO_MFMA_H2, ops.ALGO_MFMA_BF3, ops.ALGO_MFMA):
        y = ops.fused_conv(torch.from_numpy(x).permute(0, 2, 3, 1).contiguous().to(dev), torch.from_numpy(wt).to(dev), one, zero,
                           algo=algo).permute(0, 3, 1, 2).cpu().double()
        e = (y - want)[0][:, mask]
        row.append(float(e.pow(2).mean().sqrt()))
    print("2^%-3d   %.3e   %.3e   %.3e" % (k, *row))
