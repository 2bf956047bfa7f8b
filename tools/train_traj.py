#!/usr/bin/env python3
"""Loss trajectory of N full-size C3 training steps (HRNet-W32, 32 frames x 4 views, 256 x 256) with optim.Adam and with torch.optim.Adam
from the same start: an end-to-end check of the training path (P2 convs, plane-only activations, one-launch Adam, weight re-pack).
usage: train_traj.py [steps=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.optim import Adam
from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError, PoseHighResolutionNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.images(77, 32, 4, 256, 256)).reshape(128, 3, 256, 256).to(dev)
gt = torch.rand(128, 19, 64, 64, generator=torch.Generator().manual_seed(3)).to(dev) * 0.1
pv = torch.ones(128, 19, 1, 1, dtype=torch.uint8, device=dev)
loss_fn = Pose2DMeanSquaredError()
traj = {}
for name in ("mval", "torch"):
    m = PoseHighResolutionNet(19)
    sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(m._graph.param_shapes(), 0).items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    opt = Adam([{"params": m.parameters(), "lr": 1e-3}]) if name == "mval" else torch.optim.Adam([{"params": m.parameters(), "lr": 1e-3}])
    ls = []
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_fn.pose_2d_mse(m(x), gt, pv)
        loss.backward()
        opt.step()
        ls.append(float(loss.detach()))
    traj[name] = ls
    print(name, " ".join(f"{v:.5f}" for v in ls[:6]), "...", " ".join(f"{v:.5f}" for v in ls[-3:]), flush=True)
a, b = traj["mval"], traj["torch"]
print("max relative difference of the losses:", max(abs(p - q) / abs(q) for p, q in zip(a, b)), "| first/last:", a[0], a[-1])
