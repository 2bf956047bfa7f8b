#!/usr/bin/env python3
"""One HRNet-W32 training step with BatchNorm gammas spread over 2^lo .. 1 (tests/test_gpu_train.py::_spread_gammas) on every training kernel
family -- default (P2 planes), MVAL_TRAIN_P2=0 (h2), MVAL_CONV=bf3, MVAL_CONV=fp32 -- against float64 torch-CPU autograd, next to torch-CPU fp32:
which path loses what when a tensor's channels differ widely in scale.  usage: gamma_diag.py [lo_log2=-8] [n=3] [h=64] [arch=hrnet_w32] [w=h] [seed=5] [j=7]   (lo_log2 = 0: the fixture's own gammas)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import cases
from oracle import models

lo = float(sys.argv[1]) if len(sys.argv) > 1 else -8.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 64
arch = sys.argv[4] if len(sys.argv) > 4 else "hrnet_w32"
ww = int(sys.argv[5]) if len(sys.argv) > 5 else hw
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
jj = int(sys.argv[7]) if len(sys.argv) > 7 else 7
dev = torch.device("cuda:0")
c = dict(arch=arch, seed=seed, n=n, h=hw, w=ww, j=jj)
ARCH = models.HRNET_W48 if arch == "hrnet_w48" else models.HRNET_W32
rng = np.random.default_rng(1)
sd = {}
for k, v in cases.model_state_dict(c).items():
    v = torch.from_numpy(v)
    if k.endswith(".weight") and v.ndim == 1 and lo < 0:
        u = rng.uniform(lo, 0.0, size=v.shape)
        v = torch.from_numpy((np.sign(rng.standard_normal(v.shape)) * 2.0 ** u).astype(np.float32))
    sd[k] = v
x, gt, valid = cases.train_input(c)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def cpu(dt):
    sdc = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k, v in sdc.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    hm = models.hrnet_forward(sdc, torch.from_numpy(x).to(dt), ARCH, training=True)
    l = models.pose_2d_mse(hm, torch.from_numpy(gt).to(dt), torch.from_numpy(valid).reshape(hm.shape[0], -1, 1, 1))
    l.backward()
    return l.item(), {k: v.grad.numpy() for k, v in sdc.items() if v.grad is not None}


l64, g64 = cpu(torch.float64)
l32, g32 = cpu(torch.float32)
e = np.asarray([rel(g32[k], g64[k]) for k in g64])
print(f"{arch} gammas 2^{lo}..1, n={n}, {hw}x{ww}: loss64 {l64:.6f}")
print(f"{'torch-CPU fp32':28s} loss rel {abs(l32 - l64) / abs(l64):.1e}  grads vs fp64: median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} max {e.max():.2e}")
from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError

for name, env in (("default (P2)", {}), ("MVAL_TRAIN_P2=0 (h2)", {"MVAL_TRAIN_P2": "0"}), ("MVAL_CONV=bf3", {"MVAL_CONV": "bf3"}), ("MVAL_CONV=fp32", {"MVAL_CONV": "fp32"})):
    for k in ("MVAL_TRAIN_P2", "MVAL_CONV"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = cases.product_model(c)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    hm = m(torch.from_numpy(x).to(dev))
    loss = Pose2DMeanSquaredError().pose_2d_mse(hm, torch.from_numpy(gt).to(dev), torch.from_numpy(valid).reshape(hm.shape[0], -1, 1, 1).to(dev))
    loss.backward()
    errs = {k: rel(p.grad.cpu().numpy(), g64[k]) for k, p in m.named_parameters()}
    e = np.asarray(list(errs.values()))
    worst = sorted(errs, key=errs.get)[-3:]
    print(f"{name:28s} loss rel {abs(loss.item() - l64) / abs(l64):.1e}  grads vs fp64: median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} max {e.max():.2e}  worst {[(k, round(errs[k], 4)) for k in worst]}")
