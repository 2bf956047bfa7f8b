#!/usr/bin/env python3
"""Find the first operator whose split-bf16 output departs from the exact-fp32 MFMA output (eval forward,
one plan per mode, no arena re-use so that every activation survives).  usage: fwd_diag.py arch n h w"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.engine import InferencePlan

arch, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
model, _ = bench.build_model(arch, 5, dev)
x = torch.from_numpy(synth.images(3, n, 1, h, w)).reshape(n, 3, h, w).to(dev)
outs = {}
for mode in ("fp32", "bf3"):
    os.environ["MVAL_CONV"] = mode
    plan = InferencePlan(model, n, h, w, dev)
    plan.refresh_params()
    res = []
    for i, op in enumerate(plan.graph_ops):
        # run op by op; copy every output out of the (re-used) arena right away
        out = torch.empty((n, plan.out_channels) + tuple(plan.out_hw), device=dev)
        plan.run_op(i, x, out)
        torch.cuda.synchronize()
        cnt = n * (op.hout << op.up) * (op.wout << op.up) * op.cout
        res.append(plan.arena[op.out_off : op.out_off + cnt].clone() if op.out_off >= 0 else out.reshape(-1).clone())
    outs[mode] = (res, plan)
res32, res3 = outs["fp32"][0], outs["bf3"][0]
plan = outs["bf3"][1]
g = model._graph
worst = []
for i, (a, b) in enumerate(zip(res32, res3)):
    d = (a - b).abs().max().item()
    s = a.abs().max().item() + 1e-30
    op = plan.graph_ops[i]
    worst.append((d / s, i, op.algo, op.k, op.stride, op.cin, op.cout, op.hout, op.wout, op.up, g.ops[i].conv))
for r in worst:
    if r[0] > 1e-4:
        print("FIRST BAD: rel %.2e op %d algo %d k%d s%d %d->%d %dx%d up%d %s" % r)
        break
worst.sort(reverse=True)
for r in worst[:8]:
    print("rel %.2e op %d algo %d k%d s%d %d->%d %dx%d up%d %s" % r)
