#!/usr/bin/env python3
"""P2 conv kernels (csrc/conv_p2.hip): correctness against float64 torch-CPU next to the fp16-split NHWC kernel, and
launch times of the HRNet-W32 layer shapes next to that kernel's, in one process.
usage: p2_sweep.py [check|time|all] [n_images=128] [reps=50]   (MVAL_P2_TILE=ms,nt,g overrides the tile choice in a -DP2_TUNE measurement build: MVAL_BUILD_TAG=tune MVAL_EXTRA_CFLAGS=-DP2_TUNE python -m multi_view_active_learning_amd.build, then MVAL_LIB_TAG=tune)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from multi_view_active_learning_amd import _lib, ops

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
lib = _lib.lib()

# n, cin, cout, h, w, k, stride, relu, res1, res2, up, out_nchw
CHECK = [
    (3, 32, 32, 64, 64, 3, 1, True, True, False, 0, False),
    (2, 64, 64, 32, 32, 3, 1, True, False, False, 0, False),
    (2, 128, 128, 16, 16, 3, 1, True, True, False, 0, False),
    (5, 256, 256, 8, 8, 3, 1, True, True, False, 0, False),
    (2, 256, 32, 64, 64, 3, 1, True, False, False, 0, False),
    (2, 32, 64, 64, 64, 3, 2, False, True, True, 0, False),
    (2, 64, 128, 32, 32, 3, 2, True, True, True, 0, False),
    (2, 32, 32, 64, 64, 3, 2, True, False, False, 0, False),
    (3, 128, 256, 16, 16, 3, 2, True, True, False, 0, False),
    (2, 64, 32, 32, 32, 1, 1, False, True, False, 1, False),
    (2, 128, 32, 16, 16, 1, 1, True, True, False, 2, False),
    (2, 256, 32, 8, 8, 1, 1, True, True, False, 3, False),
    (2, 64, 256, 64, 64, 1, 1, True, True, False, 0, False),
    (2, 256, 64, 64, 64, 1, 1, True, False, False, 0, False),
    (2, 32, 19, 64, 64, 1, 1, False, False, False, 0, True),
    (2, 64, 64, 64, 48, 3, 1, True, False, False, 0, False),
    (1, 48, 48, 96, 72, 3, 1, True, True, False, 0, False),
    (2, 96, 96, 48, 36, 3, 1, True, True, False, 0, False),
    (1, 96, 192, 48, 36, 3, 2, True, True, True, 0, False),
    # HRNet-W48 at 384 x 288: maps 96x72 / 48x36 / 24x18 / 12x9 (partial tiles everywhere), 48-channel first branch
    (2, 192, 192, 24, 18, 3, 1, True, True, False, 0, False),
    (3, 384, 384, 12, 9, 3, 1, True, True, False, 0, False),
    (2, 48, 96, 96, 72, 3, 2, True, False, False, 0, False),
    (2, 192, 384, 24, 18, 3, 2, True, True, True, 0, False),
    (2, 48, 48, 96, 72, 3, 2, True, False, False, 0, False),
    (2, 96, 48, 48, 36, 1, 1, False, True, False, 1, False),
    (1, 192, 48, 24, 18, 1, 1, False, True, False, 2, False),
    (2, 384, 48, 12, 9, 1, 1, True, True, True, 3, False),
    (2, 384, 192, 12, 9, 1, 1, False, True, False, 1, False),
    (2, 64, 256, 96, 72, 1, 1, True, True, False, 0, False),
    (2, 256, 64, 96, 72, 1, 1, True, False, False, 0, False),
    (2, 256, 48, 96, 72, 3, 1, True, False, False, 0, False),
    (2, 256, 96, 96, 72, 3, 2, True, False, False, 0, False),
    (2, 48, 19, 96, 72, 1, 1, False, False, False, 0, True),
    (2, 64, 64, 192, 144, 3, 2, True, False, False, 0, False),
]


def ref_conv(x, w, scale, shift, stride, relu, res1, res2, up):
    y = F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2)
    y = y * scale[None, :, None, None] + shift[None, :, None, None]
    if up:
        y = F.interpolate(y, scale_factor=2**up, mode="nearest")
    if res1 is not None:
        y = y + res1
    if res2 is not None:
        y = y + res2
    return F.relu(y) if relu else y


def check():
    bad = 0
    for case in CHECK:
        n, cin, cout, h, w, k, stride, relu, r1, r2, up, out_nchw = case
        rng = np.random.default_rng(abs(hash(case)) % 2**31)
        x = torch.from_numpy(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        wt = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
        scale = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
        shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
        ho = ((h + 2 * (k // 2) - k) // stride + 1) << up
        wo = ((w + 2 * (k // 2) - k) // stride + 1) << up
        res1 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)) if r1 else None
        res2 = torch.from_numpy(rng.standard_normal((n, cout, ho, wo)).astype(np.float32)) if r2 else None
        want64 = ref_conv(x.double(), wt.double(), scale.double(), shift.double(), stride, relu,
                          None if res1 is None else res1.double(), None if res2 is None else res2.double(), up)
        nhwc = lambda t: None if t is None else t.permute(0, 2, 3, 1).contiguous().to(dev)
        try:
            got = ops.fused_conv_p2(nhwc(x), wt.to(dev), scale.to(dev), shift.to(dev), stride=stride, relu=relu, res1=nhwc(res1),
                                    res2=nhwc(res2), up=up, out_nchw=out_nchw)
        except _lib.MvalError as e:
            print("SKIP", case, e, flush=True)
            continue
        got = got.cpu() if out_nchw else got.permute(0, 3, 1, 2).cpu()
        err = (got.double() - want64).abs().max().item()
        rms = (got.double() - want64).pow(2).mean().sqrt().item()
        msg = ""
        if cin % 32 == 0 and not out_nchw:
            y = ops.fused_conv(nhwc(x), wt.to(dev), scale.to(dev), shift.to(dev), stride=stride, relu=relu, res1=nhwc(res1),
                               res2=nhwc(res2), up=up, algo=ops.ALGO_MFMA).permute(0, 3, 1, 2).cpu()
            e32 = (y.double() - want64).abs().max().item()
            r32 = (y.double() - want64).pow(2).mean().sqrt().item()
            msg = f" fp32-mfma max {e32:.2e} rms {r32:.2e} ratio rms {rms / max(r32, 1e-30):.2f}"
        if not out_nchw:
            kept = ops.fused_conv_p2.last.kept_amax().cpu()
            true = got.abs().amax(dim=(1, 2, 3))
            # the rows hold the max of the fp32 values BEFORE the 22-bit rounding of the planes
            if not torch.allclose(kept, true, rtol=1e-6, atol=0):
                msg += f" AMAX MISMATCH {kept.tolist()} vs {true.tolist()}"
                bad += 1
        ok = err <= 1e-4 * max(1.0, want64.abs().max().item())
        bad += 0 if ok else 1
        print(("ok  " if ok else "FAIL") + f" {case}: max {err:.2e} rms {rms:.2e}{msg}", flush=True)
    print("check:", "ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)


# (cin, cout, h, w, k, stride, res, up)
LAYERS = [(256, 32, 64, 64, 3, 1, 0, 0), (64, 64, 64, 64, 3, 1, 0, 0), (128, 128, 16, 16, 3, 1, 1, 0), (256, 256, 8, 8, 3, 1, 1, 0),
          (32, 32, 64, 64, 3, 1, 1, 0), (64, 64, 32, 32, 3, 1, 1, 0),
          (32, 64, 64, 64, 3, 2, 1, 0), (64, 128, 32, 32, 3, 2, 1, 0), (128, 256, 16, 16, 3, 2, 1, 0), (32, 32, 64, 64, 3, 2, 0, 0),
          (64, 256, 64, 64, 1, 1, 1, 0), (256, 64, 64, 64, 1, 1, 0, 0), (64, 32, 32, 32, 1, 1, 1, 1), (128, 32, 16, 16, 1, 1, 1, 2),
          (128, 64, 16, 16, 1, 1, 1, 1), (256, 32, 8, 8, 1, 1, 1, 3)]


def time_loop(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def timing():
    n = n_img
    for cin, cout, h, w, k, stride, res, up in LAYERS:
        ho, wo = ((h + 2 * (k // 2) - k) // stride + 1) << up, ((w + 2 * (k // 2) - k) // stride + 1) << up
        x = torch.relu(torch.randn(n, h, w, cin, device=dev))
        wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        r = torch.randn(n, ho, wo, cout, device=dev) if res else None
        fl = 2.0 * n * (ho >> up) * (wo >> up) * cin * cout * k * k
        try:
            c = ops.P2Conv(x, wt, one, zero, stride=stride, relu=True, res1=r, up=up)
            t_p2 = time_loop(c.launch)
        except _lib.MvalError as e:
            t_p2 = float("nan")
        # the NHWC fp16-split kernel on the same problem (set up once, launched through the C-ABI)
        from multi_view_active_learning_amd.engine import MvalOp, _align

        pw = ops.pack_weights(wt, ops.ALGO_MFMA_H2)
        res_off = _align(x.numel())
        out_off = res_off + _align(n * ho * wo * cout)
        amax_off = _align(out_off + n * ho * wo * cout)
        arena = torch.zeros(amax_off + _align(2 * n * 4096), device=dev)
        arena[: x.numel()] = x.reshape(-1)
        s_off = _align(pw.numel())
        params = torch.zeros(s_off + 2 * _align(cout), device=dev)
        params[: pw.numel()] = pw
        params[s_off : s_off + cout] = 1.0
        m = MvalOp()
        m.kind, m.algo = 0, ops.ALGO_MFMA_H2
        m.k, m.stride, m.pad, m.cin, m.cout = k, stride, k // 2, cin, cout
        m.hin, m.win, m.hout, m.wout = h, w, ho >> up, wo >> up
        m.up, m.relu = up, 1
        m.in_off, m.out_off, m.res1_off, m.res2_off = 0, out_off, (res_off if res else -1), -1
        m.w_off, m.scale_off, m.shift_off = 0, s_off, s_off + _align(cout)
        m.in_amax_off, m.out_amax_off = amax_off, amax_off + n * 4096
        _lib._check(lib.mval_amax(_lib._p(arena), C.c_int64(h * w * cin), C.c_int(n), C.c_void_p(arena.data_ptr() + 4 * amax_off),
                                  _lib._stream()), "mval_amax")
        t_h2 = time_loop(lambda: _lib._check(lib.mval_op_launch(C.byref(m), C.c_int(n), _lib._p(arena), _lib._p(params), C.c_void_p(0),
                                                                 C.c_void_p(0), _lib._stream()), "launch"))
        print(f"{cin:4d}->{cout:<4d} {h:3d}x{w:<3d} k{k} s{stride} up{up} res{res} n={n}: h2 {t_h2 * 1e6:7.1f} us  p2 {t_p2 * 1e6:7.1f} us  "
              f"({fl / t_p2 / 1e12:6.1f} TFLOP/s)  x{t_h2 / t_p2:.2f}", flush=True)


BLOCKS = [(3, 32, 64, 64), (2, 64, 32, 32), (2, 32, 21, 37), (1, 64, 9, 16), (5, 32, 8, 16), (2, 64, 24, 40), (2, 32, 96, 72), (1, 64, 48, 36)]


def check_blocks():
    bad = 0
    for n, c, h, w in BLOCKS:
        rng = np.random.default_rng(7 + h)
        x = torch.from_numpy(np.maximum(rng.standard_normal((n, c, h, w)), 0).astype(np.float32) * 1.5)
        ws = [torch.from_numpy((rng.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))).astype(np.float32)) for _ in range(2)]
        sc = [torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)) for _ in range(2)]
        sh = [torch.from_numpy(rng.standard_normal(c).astype(np.float32) * 0.1) for _ in range(2)]
        d = torch.float64
        mid = ref_conv(x.to(d), ws[0].to(d), sc[0].to(d), sh[0].to(d), 1, True, None, None, 0)
        want = ref_conv(mid, ws[1].to(d), sc[1].to(d), sh[1].to(d), 1, True, x.to(d), None, 0)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        got = ops.fused_basic_block_p2(xd, ws[0].to(dev), sc[0].to(dev), sh[0].to(dev), ws[1].to(dev), sc[1].to(dev), sh[1].to(dev))
        kept = ops.fused_basic_block_p2.last.kept_amax().cpu()
        got = got.permute(0, 3, 1, 2).cpu()
        ref = ops.fused_basic_block(xd, ws[0].to(dev), sc[0].to(dev), sh[0].to(dev), ws[1].to(dev), sc[1].to(dev), sh[1].to(dev)).permute(0, 3, 1, 2).cpu()
        err, rms = (got.double() - want).abs().max().item(), (got.double() - want).pow(2).mean().sqrt().item()
        rms_h2 = (ref.double() - want).pow(2).mean().sqrt().item()
        ok = err <= 1e-4 * max(1.0, want.abs().max().item()) and torch.allclose(kept, got.abs().amax(dim=(1, 2, 3)), rtol=1e-6, atol=0)
        bad += 0 if ok else 1
        print(("ok  " if ok else "FAIL") + f" block {(n, c, h, w)}: max {err:.2e} rms {rms:.2e} (h2 block rms {rms_h2:.2e})", flush=True)
    print("blocks:", "ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)


def time_blocks():
    n = n_img
    for c, h, w in ((32, 64, 64), (64, 32, 32)):
        x = torch.relu(torch.randn(n, h, w, c, device=dev))
        ws = [torch.randn(c, c, 3, 3, device=dev) * (2.0 / (c * 9)) ** 0.5 for _ in range(2)]
        one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        b = ops.P2Block(x, ws[0], one, zero, ws[1], one, zero)
        t = time_loop(b.launch)
        fl = 2 * 2.0 * n * h * w * c * c * 9
        print(f"P2 block {c}ch {h}x{w} n={n}: {t * 1e6:7.1f} us  {fl / t / 1e12:6.1f} TFLOP/s (two convs)", flush=True)


if what in ("check", "all", "blocks"):
    check_blocks()
if what in ("time", "all", "blocks"):
    time_blocks()
if what in ("check", "all"):
    check()
if what in ("time", "all"):
    timing()
