#!/usr/bin/env python3
"""Sign structure of the split-bf16 kernel's error: all-positive / all-negative products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from multi_view_active_learning_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n, cin, cout, h, w, k = 2, 384, 384, 4, 6, 3
x = np.abs(rng.standard_normal((n, cin, h, w))).astype(np.float32)
for sign, name in ((1.0, "+|w|"), (-1.0, "-|w|"), (0.0, "random w")):
    wt = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))
    wt = (np.abs(wt) * sign if sign else wt).astype(np.float32)
    xt, wtt = torch.from_numpy(x), torch.from_numpy(wt)
    want64 = F.conv2d(xt.double(), wtt.double(), None, 1, 1)
    one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    for aname, algo in (("mfma", ops.ALGO_MFMA), ("bf3", ops.ALGO_MFMA_BF3)):
        got = ops.fused_conv(xt.permute(0, 2, 3, 1).contiguous().to(dev), wtt.to(dev), one, zero, stride=1, relu=False, algo=algo)
        e = got.permute(0, 3, 1, 2).cpu().double() - want64
        print(f"{name:9s} {aname:5s} |out| mean {want64.abs().mean():.2f}  err mean-abs {e.abs().mean():.3e}  signed-mean {e.mean():+.3e}")
