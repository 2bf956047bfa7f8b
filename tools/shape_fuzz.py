#!/usr/bin/env python3
"""Eval-forward parity against the CPU oracle at unusual batch sizes and input shapes (ad-hoc robustness run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from multi_view_active_learning_amd import synth
from oracle import models
dev = torch.device("cuda:0")
for arch, n, h, w in [("hrnet_w32", 3, 384, 288), ("hrnet_w32", 2, 320, 256), ("hrnet_w32", 1, 256, 192), ("hrnet_w32", 4, 192, 256), ("hrnet_w32", 1, 512, 384),
                      ("hrnet_w32", 5, 160, 224), ("hrnet_w32", 1, 256, 256), ("hrnet_w32", 7, 96, 64), ("resnet50", 3, 224, 160),
                      ("resnet50", 1, 256, 192), ("hrnet_w48", 2, 96, 128), ("hrnet_w48", 3, 160, 96), ("hrnet_w32", 33, 64, 64),
                      # round 5: the exact-tile / small-map / flattened-1x1 / parity paths at other sizes than the BASELINE ones
                      ("hrnet_w48", 2, 384, 288), ("hrnet_w48", 2, 288, 384), ("hrnet_w48", 3, 192, 288), ("hrnet_w48", 1, 384, 192), ("hrnet_w32", 2, 384, 192),
                      ("hrnet_w32", 2, 192, 96), ("hrnet_w32", 3, 128, 128), ("resnet50", 33, 256, 192), ("resnet50", 34, 192, 256), ("resnet50", 40, 128, 96),
                      ("resnet50", 32, 384, 288), ("resnet50", 36, 160, 224)]:
    model, sd = bench.build_model(arch, 19, dev, seed=4)
    x = synth.images(11, n, 1, h, w).reshape(n, 3, h, w)
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(dev)).cpu().numpy()
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        if arch == "resnet50":
            want = models.pose_resnet_forward(sdt, torch.from_numpy(x)).numpy()
        else:
            want = models.hrnet_forward(sdt, torch.from_numpy(x), models.HRNET_W48 if arch == "hrnet_w48" else models.HRNET_W32).numpy()
    from multi_view_active_learning_amd import engine
    plan = engine._plan_for(model, torch.from_numpy(x).to(dev))
    kinds = [o.kind for o in plan.ops]
    form = f"{'P2' if plan.p2 else 'h2'} plan, {len(kinds)} launches (stem {kinds.count(6)}, bottlenecks {kinds.count(5)}, blocks {kinds.count(3)}, up-paths {kinds.count(7)})"
    err = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    am = (got.reshape(n, 19, -1).argmax(-1) == want.reshape(n, 19, -1).argmax(-1)).mean()
    print(f"{arch} n={n} {h}x{w}: shape {got.shape} rel max err {err:.2e} argmax agreement {am:.4f} | {form}", flush=True)
