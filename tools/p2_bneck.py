#!/usr/bin/env python3
"""Fused P2 Bottleneck (csrc/conv_bneck_p2.hip): correctness against float64 torch-CPU next to the three P2 conv launches it
replaces, and launch times of both.   usage: p2_bneck.py [check|time|all] [n_images=128] [reps=30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from multi_view_active_learning_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")


def make(n, cin, h, w, seed=0, with_res=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, h, w, cin, generator=g))
    convs = []
    for co, ci, k in ((64, cin, 1), (64, 64, 3), (256, 64, 1)):
        wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
        sc = 0.5 + torch.rand(co, generator=g)
        sh = 0.2 * torch.randn(co, generator=g)
        convs.append((wt, sc, sh))
    res = torch.randn(n, h, w, 256, generator=g) if (cin != 256 if with_res is None else with_res) else None
    return x, convs, res


def ref(x, convs, res, dt=torch.float64):
    t = x.permute(0, 3, 1, 2).to(dt)
    r = t if res is None else res.permute(0, 3, 1, 2).to(dt)
    for i, (wt, sc, sh) in enumerate(convs):
        t = F.conv2d(t, wt.to(dt), padding=wt.shape[-1] // 2) * sc.to(dt).view(1, -1, 1, 1) + sh.to(dt).view(1, -1, 1, 1)
        if i == 2:
            t = t + r
        t = torch.relu(t)
    return t.permute(0, 2, 3, 1)


def chain(x, convs, res):
    """the same block as three P2 conv launches"""
    d = lambda t: t.to(dev)
    (w1, s1, h1), (w2, s2, h2), (w3, s3, h3) = convs
    a = ops.fused_conv_p2(d(x), d(w1), d(s1), d(h1), relu=True)
    b = ops.fused_conv_p2(a, d(w2), d(s2), d(h2), relu=True)
    return ops.fused_conv_p2(b, d(w3), d(s3), d(h3), relu=True, res1=d(x if res is None else res))


def check():
    bad = 0
    for n, cin, h, w, wr in ((2, 256, 64, 64, None), (2, 64, 64, 64, None), (3, 256, 21, 37, None), (1, 64, 9, 16, None), (2, 256, 96, 72, None),
                             (2, 256, 64, 48, True), (5, 64, 8, 16, None)):
        x, convs, res = make(n, cin, h, w, seed=n + cin + h, with_res=wr)
        want = ref(x, convs, res)
        got = ops.fused_bottleneck_p2(x.to(dev), [tuple(t.to(dev) for t in c) for c in convs], None if res is None else res.to(dev)).cpu().double()
        ch = chain(x, convs, res).cpu().double()
        kept = ops.fused_bottleneck_p2.last.kept_amax().cpu()
        e = lambda y: ((y - want).abs().max().item(), (y - want).pow(2).mean().sqrt().item())
        (em, er), (cm, cr) = e(got), e(ch)
        amax_ok = torch.allclose(kept, got.float().abs().amax(dim=(1, 2, 3)), rtol=2.0**-21, atol=0)
        ok = em <= 2.5 * cm + 1e-6 and er <= 1.25 * cr + 1e-8 and amax_ok
        bad += not ok
        print(f"n{n} cin{cin} {h}x{w} res={'x' if res is None else 'r'}: fused max {em:.3e} rms {er:.3e} | 3 launches max {cm:.3e} rms {cr:.3e} | "
              f"|want| max {want.abs().max():.2f} amax {'ok' if amax_ok else 'BAD'} {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def time_loop(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def timing():
    for cin, h, w in ((256, 64, 64), (64, 64, 64), (256, 96, 72)):
        n = n_img
        x, convs, res = make(n, cin, h, w)
        d = lambda t: t.to(dev)
        cv = [tuple(d(t) for t in c) for c in convs]
        b = ops.P2Bneck(d(x), cv, None if res is None else d(res))
        t_f = time_loop(b.launch)
        (w1, s1, h1), (w2, s2, h2), (w3, s3, h3) = cv
        mid = torch.relu(torch.randn(n, h, w, 64, device=dev))
        c1 = ops.P2Conv(d(x), w1, s1, h1, relu=True)
        c2 = ops.P2Conv(mid, w2, s2, h2, relu=True)
        c3 = ops.P2Conv(mid, w3, s3, h3, relu=True, res1=d(x if res is None else res))
        ts = [time_loop(c.launch) for c in (c1, c2, c3)]
        fl = 2.0 * n * h * w * (cin * 64 + 64 * 64 * 9 + 64 * 256)
        print(f"cin{cin} {h}x{w} n={n}: fused {t_f * 1e6:7.1f} us ({fl / t_f / 1e12:6.1f} TFLOP/s) | three launches {sum(ts) * 1e6:7.1f} us "
              f"({' + '.join('%.1f' % (t * 1e6) for t in ts)})  x{sum(ts) / t_f:.2f}", flush=True)


rc = 0
if what in ("check", "all"):
    rc = check()
if what in ("time", "all"):
    timing()
sys.exit(1 if rc else 0)
