#!/usr/bin/env python3
"""Latency of the scoring path at the reference's DEFAULT batch sizes (AL.INFERENCE.BATCH_SIZE = 2 frames).
usage: small_batch.py [arch=hrnet_w32] [frames=2] [views=4] [H=256] [W=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.utils.triangulation import triangulate_batch

arch = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
v = int(sys.argv[3]) if len(sys.argv) > 3 else 4
h = int(sys.argv[4]) if len(sys.argv) > 4 else 256
w = int(sys.argv[5]) if len(sys.argv) > 5 else 256
dev = torch.device("cuda:0")
model, _ = bench.build_model(arch, 19, dev)
x = torch.from_numpy(synth.images(0, frames, v, h, w)).to(dev).reshape(frames * v, 3, h, w)
proj = torch.from_numpy(np.stack([synth.ring_cameras(v, h, w, seed=s) for s in range(frames)])).to(dev)
valid = torch.ones(frames, 19, dtype=torch.uint8, device=dev)


def step():
    hm = model(x).reshape(frames, v, 19, h // 4, w // 4)
    return triangulate_batch(hm, proj, 4, valid)


with torch.no_grad():
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(x)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / n
print(f"{arch} {frames} frames x {v} views {h}x{w}: step {dt * 1e3:.3f} ms ({frames * v / dt:.0f} frames*views/s), network alone {dtm * 1e3:.3f} ms")
