#!/usr/bin/env python3
"""Time the 1x1 layer shapes (bottleneck blocks of HRNet's layer1 / PoseResNet-50) through the C-ABI.
usage: conv1x1_sweep.py [n_images=128] [reps=50]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multi_view_active_learning_amd import _lib, ops
from multi_view_active_learning_amd.engine import MvalOp, _align

algo = ops.ALGO_MFMA_BF3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
lib = _lib.lib()

# (cin, cout, h, w, residual)
LAYERS = [(64, 256, 64, 64, 1), (64, 256, 64, 64, 0), (256, 64, 64, 64, 0), (64, 64, 64, 64, 0),
          (256, 64, 64, 48, 0), (64, 256, 64, 48, 1), (512, 128, 32, 24, 0), (128, 512, 32, 24, 1),
          (1024, 256, 16, 12, 0), (256, 1024, 16, 12, 1), (2048, 512, 8, 6, 0), (512, 2048, 8, 6, 1)]


def bench(cin, cout, h, w, res):
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, 1, 1, device=dev) * (2.0 / cin) ** 0.5
    pw = ops.pack_weights(wt, algo)
    res_off = _align(x.numel())
    out_off = res_off + _align(n * h * w * cout)
    arena = torch.zeros(out_off + n * h * w * cout, device=dev)
    arena[: x.numel()] = x.reshape(-1)
    arena[res_off : res_off + n * h * w * cout] = torch.randn(n * h * w * cout, device=dev)
    s_off = _align(pw.numel())
    params = torch.zeros(s_off + 2 * _align(cout), device=dev)
    params[: pw.numel()] = pw
    params[s_off : s_off + cout] = 1.0
    m = MvalOp()
    m.kind, m.algo = 0, algo
    m.k, m.stride, m.pad, m.cin, m.cout = 1, 1, 0, cin, cout
    m.hin, m.win, m.hout, m.wout = h, w, h, w
    m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
    m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, (res_off if res else -1), -1
    m.w_off, m.scale_off, m.shift_off = 0, s_off, s_off + _align(cout)

    def run():
        _lib._check(lib.mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0),
                                       C.c_void_p(0), _lib._stream()), "launch")

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * n * h * w * cin * cout
    byt = 4.0 * n * h * w * (cin + cout * (1 + res))
    out = arena[out_off:].reshape(n, h, w, cout)
    ref = torch.relu(x.reshape(-1, cin)[:4096] @ wt.reshape(cout, cin).t()
                     + (arena[res_off : res_off + n * h * w * cout].reshape(-1, cout)[:4096] if res else 0))
    err = float((out.reshape(-1, cout)[:4096] - ref).abs().max())
    print(f"{cin:4d}->{cout:4d} {h}x{w} res={res} n={n}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s  {byt / dt / 1e9:6.0f} GB/s  err {err:.1e}",
          flush=True)


for layer in LAYERS:
    bench(*layer)
