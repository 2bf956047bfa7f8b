#!/usr/bin/env python3
"""Per-operator breakdown of one HRNet-W32 training step (C3 shapes): hipEvents around every launch group of
mval_train_forward / mval_train_backward, grouped by operator shape x kernel family.
usage: train_op_times.py [n_images=128] [arch=hrnet_w32]"""
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib, synth
from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError, PoseHighResolutionNet, hrnet_w48

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
arch = sys.argv[2] if len(sys.argv) > 2 else "hrnet_w32"
dev = torch.device("cuda:0")
m = PoseHighResolutionNet(19) if arch == "hrnet_w32" else PoseHighResolutionNet(19, hrnet_cfg=hrnet_w48())
sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(m._graph.param_shapes(), 0).items()}
m.load_state_dict(sd, strict=True)
m = m.to(dev).train()
h, w = (256, 256) if arch == "hrnet_w32" else (384, 288)
x = torch.randn(n, 3, h, w, device=dev)
gt = torch.rand(n, 19, h // 4, w // 4, device=dev)
pv = torch.ones(n, 19, 1, 1, dtype=torch.uint8, device=dev)
loss_fn = Pose2DMeanSquaredError()


def step():
    m.zero_grad()
    loss_fn.pose_2d_mse(m(x), gt, pv).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
plan = next(iter(m._train_plans.values()))
nops = len(plan.ops)
fam = (C.c_float * 6)()
per = (C.c_float * (nops * 6))()
lib = _lib.lib()
reps = 3
lib.mval_train_timing_ops(fam, per, C.c_int(nops))
try:
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
finally:
    lib.mval_train_timing(None)
per = np.frombuffer(per, dtype=np.float32).reshape(nops, 6) / reps
names = ["conv_fwd", "bn_stats", "bn_apply", "bn_bwd", "wgrad", "dgrad"]
print("families (ms/step):", {k: round(float(v) / reps, 3) for k, v in zip(names, fam)}, "sum", round(sum(fam) / reps, 3))
g = defaultdict(lambda: np.zeros(7))
for t, row in zip(plan.ops, per):
    o = t.op
    key = (o.k, o.stride, o.cin, o.cout, o.hout, o.wout, o.up, int(o.res1_off >= 0) + int(o.res2_off >= 0), o.algo, t.dgrad_algo)
    g[key][:6] += row
    g[key][6] += 1
print(f"{'k s cin->cout  HxW up res algo/dg':40s} {'cnt':>4s} " + " ".join(f"{x:>9s}" for x in names) + "   (us per op)")
for key, v in sorted(g.items(), key=lambda kv: -kv[1][:6].sum()):
    k, s, ci, co, ho, wo, up, nres, algo, dg = key
    cnt = v[6]
    print(f"k{k}s{s} {ci:4d}->{co:<4d} {ho:3d}x{wo:<3d} up{up} res{nres} a{algo}/{dg:<2d}".ljust(40) + f" {int(cnt):4d} "
          + " ".join(f"{x / cnt * 1e3:9.1f}" for x in v[:6]) + f"   total {v[:6].sum():7.3f} ms")
