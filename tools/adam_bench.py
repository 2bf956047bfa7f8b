#!/usr/bin/env python3
"""optim.Adam (one launch) next to torch.optim.Adam (foreach / fused) on HRNet-W32's parameter list: ms per step (hipEvents)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_view_active_learning_amd.optim import Adam
from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet

dev = torch.device("cuda:0")
m = PoseHighResolutionNet(19).to(dev)
ps = list(m.parameters())
total = sum(p.numel() for p in ps)
flat = torch.randn(total + 4 * len(ps), device=dev)
def set_grads():
    off = 0
    for p in ps:
        p.grad = flat[off : off + p.numel()].view_as(p)
        off += (p.numel() + 3) & ~3
for name, opt in (("mval", Adam([{"params": ps, "lr": 1e-3}])), ("torch foreach", torch.optim.Adam([{"params": ps, "lr": 1e-3}])),
                  ("torch fused", torch.optim.Adam([{"params": ps, "lr": 1e-3}], fused=True))):
    set_grads()
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        opt.step()
    e1.record()
    host = (time.perf_counter() - t0) / 20 * 1e3
    e1.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / 20:.3f} ms per step on the device, {host:.3f} ms of host time; {len(ps)} tensors, {total / 1e6:.1f} M parameters")
