#!/usr/bin/env python3
"""Micro-benchmark of one fused conv operator through the C-ABI (for rocprofv3 --pmc runs).
usage: conv_bench.py cin cout h w k stride algo(mfma|bf3|direct) n_images [reps] [res]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch

from multi_view_active_learning_amd import _lib, ops
from multi_view_active_learning_amd.engine import MvalOp, _align

cin, cout, h, w, k, stride = map(int, sys.argv[1:7])
algo = {"mfma": ops.ALGO_MFMA, "bf3": ops.ALGO_MFMA_BF3, "direct": ops.ALGO_DIRECT}[sys.argv[7]]
n = int(sys.argv[8])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 50
res = int(sys.argv[10]) if len(sys.argv) > 10 else 1
dev = torch.device("cuda:0")
ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
x = torch.randn(n, h, w, cin, device=dev)
wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
pw = ops.pack_weights(wt, algo)
in_off, res_off = 0, _align(x.numel())
out_off = res_off + _align(n * ho * wo * cout)
arena = torch.zeros(out_off + n * ho * wo * cout, device=dev)
arena[: x.numel()] = x.reshape(-1)
arena[res_off : res_off + n * ho * wo * cout] = torch.randn(n * ho * wo * cout, device=dev)
s_off = _align(pw.numel())
params = torch.zeros(s_off + 2 * _align(cout), device=dev)
params[: pw.numel()] = pw
params[s_off : s_off + cout] = 1.0
m = MvalOp()
m.kind, m.algo = 0, algo
m.k, m.stride, m.pad, m.cin, m.cout = k, stride, k // 2, cin, cout
m.hin, m.win, m.hout, m.wout = h, w, ho, wo
m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
m.in_off, m.out_off, m.res1_off, m.res2_off = in_off, out_off, (res_off if res else -1), -1
m.w_off, m.scale_off, m.shift_off = 0, s_off, s_off + _align(cout)
lib = _lib.lib()


def run():
    _lib._check(lib.mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0), C.c_void_p(0),
                                   _lib._stream()), "launch")


for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * n * ho * wo * cin * cout * k * k
print(f"{sys.argv[7]} {cin}->{cout} {h}x{w} k{k}s{stride} n={n}: {dt * 1e6:.1f} us  {fl / dt / 1e12:.1f} TFLOP/s")
