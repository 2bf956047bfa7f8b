#!/usr/bin/env python3
"""Per-op timing of one network forward (hipEvents around every launch, mval_net_forward_timed),
grouped by op geometry.  Usage (GPU box): python tools/op_profile.py [arch] [n_images] [H] [W]"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.engine import _plan_for

arch = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
h = int(sys.argv[3]) if len(sys.argv) > 3 else 256
w = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
model, _ = bench.build_model(arch, 19, dev)
x = torch.from_numpy(synth.images(0, n, 1, h, w)).reshape(n, 3, h, w).to(dev)
plan = _plan_for(model, x)
with torch.no_grad():
    plan.forward(x)
    acc = None
    for _ in range(5):
        _, ms, fl = plan.forward_timed(x)
        acc = ms if acc is None else acc + ms
ms = acc / 5
groups = defaultdict(lambda: [0, 0.0, 0.0])
for o, t, f in zip(plan.ops, ms, fl):
    key = (o.kind, o.algo, o.k, o.stride, o.cin, o.cout, o.hout, o.wout, o.up, int(o.res1_off >= 0) + int(o.res2_off >= 0))
    g = groups[key]
    g[0] += 1
    g[1] += t
    g[2] += f
print(f"total {ms.sum():.3f} ms, {fl.sum() / ms.sum() / 1e9:.1f} TFLOP/s")
print("kind algo k s cin cout hout wout up res | count  total_ms  avg_us   TFLOP/s  GB/s(min traffic)")
for key, (c, t, f) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    kind, algo, k, s, cin, cout, ho, wo, up, nres = key
    byt = 4.0 * n * (ho * s * wo * s * cin + (ho << up) * (wo << up) * cout * (1 + nres)) * c
    print(f"{kind} {algo} {k} {s} {cin:4d} {cout:4d} {ho:4d} {wo:4d} {up} {nres} | {c:4d} {t:9.3f} {t / c * 1e3:8.1f} {f / t / 1e9:9.1f} {byt / t / 1e6:9.0f}")
