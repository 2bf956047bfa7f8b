#!/usr/bin/env python3
"""Train-mode forward, split-bf16 vs exact-fp32 conv kernels: first operator whose raw conv output (z) or
activation departs.  usage: train_fwd_diag.py arch n h w [mode=bf3|h2|p2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.engine_train import TrainPlan

arch, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
other = sys.argv[5] if len(sys.argv) > 5 else "bf3"
dev = torch.device("cuda:0")
model, _ = bench.build_model(arch, 5, dev)
model.train()
x = torch.from_numpy(synth.images(3, n, 1, h, w)).reshape(n, 3, h, w).to(dev)
arenas = {}
for mode in ("fp32", other):
    os.environ["MVAL_CONV"] = mode
    plan = TrainPlan(model, n, h, w, dev)
    with torch.no_grad():
        out = plan.forward(x)
    torch.cuda.synchronize()
    arenas[mode] = (plan.arena.clone(), plan, out.clone())
a32, p32, o32 = arenas["fp32"]
a3, p3, o3 = arenas[other]
g = model._graph
print("output rel diff", ((o32 - o3).abs().max() / o32.abs().max()).item())
rows = []
for i, (t, op) in enumerate(zip(p3.ops, g.ops)):
    m = t.op
    cnt = n * m.hout * m.wout * m.cout
    if t.z_off >= 0:
        z32, z3 = a32[t.z_off : t.z_off + cnt], a3[t.z_off : t.z_off + cnt]
        zz32, zz3 = z32.reshape(-1, m.cout).double(), z3.reshape(-1, m.cout).double()
        perc = ((zz32 - zz3).abs().max(0).values / (zz32.std(0) + 1e-30))  # per channel, in units of its std
        rows.append((perc.max().item(), i, "z ch%d" % int(perc.argmax()), m.algo, m.k, m.stride, m.cin, m.cout, m.hout, m.wout, op.conv + (" [fwd_p2]" if t.fwd_p2 else "")))
for r in rows:
    if r[0] > 1e-3:
        print("FIRST BAD rel %.2e op %d %s algo %d k%d s%d %d->%d %dx%d %s" % r)
        break
rows.sort(reverse=True)
for r in rows[:6]:
    print("rel %.2e op %d %s algo %d k%d s%d %d->%d %dx%d %s" % r)
