#!/usr/bin/env python3
"""Fused P2 stem (csrc/conv_stem_p2.hip): launch time next to the three launches it replaces (fp32 stem kernel, format change,
P2 stride-2 conv).   usage: p2_stem.py [n_images=128] [h=256] [w=256] [reps=30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multi_view_active_learning_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda:0")
x = torch.randn(n, 3, h, w, device=dev)
w1 = torch.randn(64, 3, 3, 3, device=dev) * (2.0 / 27) ** 0.5
w2 = torch.randn(64, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
b = ops.P2Stem(x, w1, one, zero, w2, one, zero)


def time_loop(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_f = time_loop(b.launch)
y1 = torch.relu(torch.randn(n, h // 2, w // 2, 64, device=dev))
c2 = ops.P2Conv(y1, w2, one, zero, stride=2, relu=True)
t_c2 = time_loop(c2.launch)
t_s = time_loop(lambda: ops.fused_conv(x, w1, one, zero, stride=2, relu=True, algo=ops.ALGO_DIRECT, in_nchw=True))
fl = 2.0 * n * ((h // 2) * (w // 2) * 27 * 64 + (h // 4) * (w // 4) * 576 * 64)
print(f"stem {h}x{w} n={n}: fused {t_f * 1e6:7.1f} us ({fl / t_f / 1e12:6.1f} TFLOP/s) | P2 conv2 alone {t_c2 * 1e6:7.1f} us, fp32 stem via ops.fused_conv (with its "
      f"host-side setup) {t_s * 1e6:7.1f} us", flush=True)
