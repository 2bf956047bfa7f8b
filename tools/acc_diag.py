#!/usr/bin/env python3
"""Accuracy of the two MFMA conv kernels against a float64 reference on network-like data
(post-ReLU inputs, small maps, deep K).  usage: acc_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from multi_view_active_learning_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for (n, cin, cout, h, w, k, relu_in) in [(2, 384, 384, 2, 3, 3, True), (2, 384, 384, 2, 3, 3, False), (2, 192, 192, 4, 6, 3, True),
                                         (8, 256, 256, 8, 8, 3, True), (2, 48, 48, 16, 24, 3, True), (2, 96, 96, 8, 12, 3, True)]:
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    if relu_in:
        x = np.maximum(x + 0.3, 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)
    xt, wtt = torch.from_numpy(x), torch.from_numpy(wt)
    want64 = F.conv2d(xt.double(), wtt.double(), None, 1, k // 2)
    cpu32 = F.conv2d(xt, wtt, None, 1, k // 2).double()
    one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    res = {}
    for name, algo in (("mfma", ops.ALGO_MFMA), ("bf3", ops.ALGO_MFMA_BF3)):
        got = ops.fused_conv(xt.permute(0, 2, 3, 1).contiguous().to(dev), wtt.to(dev), one, zero, stride=1, relu=False, algo=algo)
        res[name] = got.permute(0, 3, 1, 2).cpu().double()
    def stats(a):
        e = (a - want64).abs()
        return e.max().item(), e.mean().item(), (a - want64).mean().item()
    print(f"{cin}->{cout} {h}x{w} relu_in={relu_in}: |out| max {want64.abs().max():.2f}")
    for name, a in (("cpu32", cpu32), ("mfma", res["mfma"]), ("bf3", res["bf3"])):
        mx, mean, bias = stats(a)
        print(f"   {name:6s} max {mx:.3e} mean {mean:.3e} signed-mean {bias:+.3e}")
