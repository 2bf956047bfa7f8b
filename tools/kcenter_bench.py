#!/usr/bin/env python3
"""Core-set (k-center) selection timing at the BASELINE pool size (50 000 + 200 rows, D = 57,
100 picks) + HBM roofline of the per-step kernel.  GPU box: python tools/kcenter_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd import _lib

dev = torch.device("cuda:0")
n_pool, n_lab, j, picks = 50000, 200, 19, 100
rng = np.random.default_rng(0)
feat = torch.from_numpy(rng.standard_normal((n_pool + n_lab, 3 * j)) * 300.0).to(dev)
lab = torch.arange(n_pool, n_pool + n_lab, device=dev)
for _ in range(2):
    p, md = _lib.kcenter_select(feat, lab, picks)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    p, md = _lib.kcenter_select(feat, lab, picks)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / reps
# per-step: separate timing with 0 picks (transpose + init only)
t0 = time.perf_counter()
for _ in range(reps):
    _lib.kcenter_select(feat, lab, 0)
torch.cuda.synchronize()
init = (time.perf_counter() - t0) / reps
step = (total - init) / picks
n_obs, d = feat.shape
bytes_step = n_obs * d * 8 + 2 * n_obs * 8 + n_obs * 8  # features + min_d read/write + norms
print(json.dumps(dict(n_obs=n_obs, D=d, picks=picks, select_batch_ms=round(total * 1e3, 3), init_ms=round(init * 1e3, 3),
                      step_us=round(step * 1e6, 2), bytes_per_step=bytes_step,
                      achieved_GBps=round(bytes_step / step / 1e9, 1), hbm_peak_GBps=8000,
                      frac=round(bytes_step / step / 8e12, 4),
                      note="22.9 MB table is L2/MALL resident; a step is launch-latency bound at this size")))
