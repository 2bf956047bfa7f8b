#!/usr/bin/env python3
"""Time the device input pipeline (crop + LANCZOS resize + normalise, GT heat-maps) at the C2 batch shape
against PIL on the host.  usage: preprocess_bench.py [views=128] [raw=720] [crop=512] [out=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from multi_view_active_learning_amd.utils import preprocess

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 128
raw = int(sys.argv[2]) if len(sys.argv) > 2 else 720
crop = int(sys.argv[3]) if len(sys.argv) > 3 else 512
out = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
host = [rng.integers(0, 256, size=(raw, raw * 16 // 9, 3), dtype=np.uint8) for _ in range(8)]
imgs = [torch.from_numpy(host[i % 8]).to(dev) for i in range(nv)]
boxes = [(100 + (i % 7) * 5, 60 + (i % 5) * 7, 100 + (i % 7) * 5 + crop, 60 + (i % 5) * 7 + crop) for i in range(nv)]
pt = torch.from_numpy(rng.uniform(0, out // 4, size=(nv, 19, 2))).to(dev)
for _ in range(3):
    preprocess.resize_views(imgs, boxes, out, out)
    preprocess.gt_heatmaps(pt, 1.0, out // 4, out // 4)
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    o = preprocess.resize_views(imgs, boxes, out, out)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    h = preprocess.gt_heatmaps(pt, 1.0, out // 4, out // 4)
torch.cuda.synchronize()
t2 = time.perf_counter()
dt_r, dt_h = (t1 - t0) / reps, (t2 - t1) / reps
byt = nv * (crop * crop * 3 + crop * out * 3 * 2 + out * out * 12)
print(f"resize_views: {nv} views {crop}x{crop} -> {out}x{out}: {dt_r * 1e3:.3f} ms ({nv / dt_r:.0f} views/s, {byt / dt_r / 1e9:.0f} GB/s of "
      f"crop-in + temp + out bytes; includes the per-call host descriptor build)")
print(f"gt_heatmaps: {nv} x 19 maps {out // 4}x{out // 4}: {dt_h * 1e3:.3f} ms")
from PIL import Image

t0 = time.perf_counter()
n_cpu = 16
for i in range(n_cpu):
    b = boxes[i]
    im = Image.fromarray(host[i % 8][..., ::-1]).crop(b).resize((out, out), resample=Image.LANCZOS)
    _ = (np.asarray(im) / 255.0 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225])
dt_c = (time.perf_counter() - t0) / n_cpu
print(f"PIL on one host core: {dt_c * 1e3:.2f} ms / view ({1 / dt_c:.0f} views/s)")
