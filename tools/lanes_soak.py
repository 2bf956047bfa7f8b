#!/usr/bin/env python3
"""Soak of the training lanes (event-ordered gradient slots, join-less backward): the SAME full-size C3 step `reps` times from the same
state -- every parameter gradient, the loss and the running statistics must repeat bit for bit (a missing dependency between lanes
shows up as a sporadic difference), and equal the one-stream step's (MVAL_TRAIN_LANES=0).  usage: lanes_soak.py [reps=30] [arch=hrnet_w32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError, PoseHighResolutionNet, hrnet_w48

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
arch = sys.argv[2] if len(sys.argv) > 2 else "hrnet_w32"
dev = torch.device("cuda:0")
n, h, w = (128, 256, 256) if arch == "hrnet_w32" else (32, 384, 288)


def build():
    m = PoseHighResolutionNet(19) if arch == "hrnet_w32" else PoseHighResolutionNet(19, hrnet_cfg=hrnet_w48())
    sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(m._graph.param_shapes(), 0).items()}
    m.load_state_dict(sd, strict=True)
    return m.to(dev).train(), sd


x = torch.randn(n, 3, h, w, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
gt = torch.rand(n, 19, h // 4, w // 4, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
pv = torch.ones(n, 19, 1, 1, dtype=torch.uint8, device=dev)
loss_fn = Pose2DMeanSquaredError()


def step(m, sd):
    m.load_state_dict(sd, strict=True)
    m.zero_grad()
    loss = loss_fn.pose_2d_mse(m(x), gt, pv)
    loss.backward()
    return loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters()], [b.detach().clone() for k, b in m.named_buffers() if "running" in k]


m, sd = build()
ref = step(m, sd)
bad = 0
for r in range(reps):
    cur = step(m, sd)
    ok = torch.equal(cur[0], ref[0]) and all(torch.equal(a, b) for a, b in zip(cur[1], ref[1])) and all(torch.equal(a, b) for a, b in zip(cur[2], ref[2]))
    bad += not ok
    if not ok:
        print(f"repeat {r}: DIFFERS from the first step", flush=True)
plan = next(iter(m._train_plans.values()))
print(f"{arch} {n} x {h}x{w}, {plan.n_lanes} lanes: {reps} repeats, {bad} differ; loss {float(ref[0]):.6f}", flush=True)
os.environ["MVAL_TRAIN_LANES"] = "0"
m0, sd0 = build()
one = step(m0, sd0)
same = torch.equal(one[0], ref[0]) and all(torch.equal(a, b) for a, b in zip(one[1], ref[1])) and all(torch.equal(a, b) for a, b in zip(one[2], ref[2]))
print(f"one-stream step (MVAL_TRAIN_LANES=0): {'identical' if same else 'DIFFERENT'}", flush=True)
sys.exit(1 if (bad or not same) else 0)
