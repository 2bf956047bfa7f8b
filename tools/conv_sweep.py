#!/usr/bin/env python3
"""Time the HRNet-W32 3x3 layer shapes through the C-ABI in one process (kernel-variant A/B runs:
one run per kernel family).
usage: conv_sweep.py [algo=bf3|h2|mfma] [n_images=128] [reps=100]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multi_view_active_learning_amd import _lib, ops
from multi_view_active_learning_amd.engine import MvalOp, _align

algo_name = sys.argv[1] if len(sys.argv) > 1 else "bf3"
algo = {"mfma": ops.ALGO_MFMA, "bf3": ops.ALGO_MFMA_BF3, "h2": ops.ALGO_MFMA_H2}[algo_name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
lib = _lib.lib()

# (cin, cout, h, w, stride)
LAYERS = [(32, 32, 64, 64, 1), (64, 64, 64, 64, 1), (64, 64, 32, 32, 1), (128, 128, 16, 16, 1), (256, 256, 8, 8, 1),
          (32, 64, 64, 64, 2), (64, 128, 32, 32, 2), (128, 256, 16, 16, 2)]


def bench(cin, cout, h, w, stride, k=3):
    ho, wo = (h + 2 - k) // stride + 1, (w + 2 - k) // stride + 1
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
    pw = ops.pack_weights(wt, algo)
    res_off = _align(x.numel())
    out_off = res_off + _align(n * ho * wo * cout)
    amax_off = _align(out_off + n * ho * wo * cout)  # per-image max |x| slots (input, output)
    arena = torch.zeros(amax_off + _align(2 * n * 4096), device=dev)
    arena[: x.numel()] = x.reshape(-1)
    arena[res_off : res_off + n * ho * wo * cout] = torch.randn(n * ho * wo * cout, device=dev)
    s_off = _align(pw.numel())
    params = torch.zeros(s_off + 2 * _align(cout), device=dev)
    params[: pw.numel()] = pw
    params[s_off : s_off + cout] = 1.0
    m = MvalOp()
    m.kind, m.algo = 0, algo
    m.k, m.stride, m.pad, m.cin, m.cout = k, stride, 1, cin, cout
    m.hin, m.win, m.hout, m.wout = h, w, ho, wo
    m.up, m.relu, m.in_nchw, m.out_nchw = 0, 1, 0, 0
    m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, res_off, -1
    m.w_off, m.scale_off, m.shift_off = 0, s_off, s_off + _align(cout)
    m.in_amax_off, m.out_amax_off = amax_off, amax_off + n * 4096
    _lib._check(lib.mval_amax(_lib._p(arena), C.c_int64(h * w * cin), C.c_int(n), C.c_void_p(arena.data_ptr() + 4 * amax_off),
                              _lib._stream()), "mval_amax")

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
    fl = 2.0 * n * ho * wo * cin * cout * k * k
    print(f"{algo_name} {cin}->{cout} {h}x{w} s{stride} n={n}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s", flush=True)


for layer in LAYERS:
    bench(*layer)


def bench_block(c, h, w):
    """A whole BasicBlock (MVAL_OP_BLOCK) on n images of c x h x w, next to the same block as two h2 launches."""
    x = torch.relu(torch.randn(n, h, w, c, device=dev))
    ws = [torch.randn(c, c, 3, 3, device=dev) * (2.0 / (c * 9)) ** 0.5 for _ in range(2)]
    one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)

    def run():
        return ops.fused_basic_block(x, ws[0], one, zero, ws[1], one, zero)

    # ops.fused_basic_block re-packs weights and re-builds its arena per call: time the launch alone through events
    for _ in range(3):
        run()
    # build once, launch many
    pw = [ops.pack_weights(wt, ops.ALGO_MFMA_H2) for wt in ws]
    out_off = _align(x.numel())
    amax_off = _align(out_off + x.numel())
    arena = torch.zeros(amax_off + _align(2 * n * 4096), device=dev)
    arena[: x.numel()] = x.reshape(-1)
    offs, top = [], 0
    for t in (pw[0], one, zero, pw[1], one, zero):
        offs.append(top)
        top += _align(t.numel())
    params = torch.zeros(top, device=dev)
    for o, t in zip(offs, (pw[0], one, zero, pw[1], one, zero)):
        params[o : o + t.numel()] = t
    m = MvalOp()
    m.kind, m.algo = 3, ops.ALGO_MFMA_H2
    m.k, m.stride, m.pad, m.cin, m.cout = 3, 1, 1, c, c
    m.hin, m.win, m.hout, m.wout = h, w, h, w
    m.up, m.relu = 0, 1
    m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, 0, -1
    m.w_off, m.scale_off, m.shift_off, m.w2_off, m.scale2_off, m.shift2_off = offs
    m.in_amax_off, m.out_amax_off = amax_off, amax_off + n * 4096
    _lib._check(lib.mval_amax(_lib._p(arena), C.c_int64(h * w * c), C.c_int(n), C.c_void_p(arena.data_ptr() + 4 * amax_off),
                              _lib._stream()), "mval_amax")

    def launch():
        _lib._check(lib.mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0),
                                       C.c_void_p(0), _lib._stream()), "launch")

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fl = 2 * 2.0 * n * h * w * c * c * 9
    print(f"block {c}ch {h}x{w} n={n}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s (two convs)", flush=True)


if algo_name == "h2":
    bench_block(32, 64, 64)
    bench_block(64, 32, 32)
