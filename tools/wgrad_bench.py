#!/usr/bin/env python3
"""Time mval_conv_wgrad on the HRNet-W32 layer shapes (one process; kernel knobs via the environment).
usage: wgrad_bench.py [n_images=128] [reps=50]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multi_view_active_learning_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
lib, p = _lib.lib(), _lib._p
lib.mval_conv_wgrad_workspace_floats.restype = C.c_size_t
# (cin, cout, h, w, k, stride)
LAYERS = [(32, 32, 64, 64, 3, 1), (64, 64, 32, 32, 3, 1), (128, 128, 16, 16, 3, 1), (256, 256, 8, 8, 3, 1),
          (64, 64, 64, 64, 3, 1), (256, 64, 64, 64, 1, 1), (64, 256, 64, 64, 1, 1), (32, 64, 64, 64, 3, 2)]
for cin, cout, h, w, k, s in LAYERS:
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    x = torch.randn(n, h, w, cin, device=dev)
    dz = torch.randn(n, ho, wo, cout, device=dev)
    ws = torch.empty(int(lib.mval_conv_wgrad_workspace_floats(C.c_int(cin), C.c_int(cout), C.c_int(k))) + 64, device=dev)
    dw = torch.empty((cout, cin, k, k), device=dev)

    def run():
        _lib._check(lib.mval_conv_wgrad(p(x), p(dz), p(dw), p(ws), C.c_int(n), C.c_int(h), C.c_int(w), C.c_int(cin),
                                        C.c_int(ho), C.c_int(wo), C.c_int(cout), C.c_int(k), C.c_int(s), C.c_int(k // 2),
                                        C.c_int(0), _lib._stream()), "wgrad")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * n * ho * wo * cin * cout * k * k
    print(f"wgrad {cin}->{cout} {h}x{w} k{k}s{s} n={n}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s (incl. slab reduce)", flush=True)
