#!/usr/bin/env python3
"""Per-op times of an inference plan (hipEvents around every launch), grouped by operator shape.
usage: op_times.py [n_images=128] [arch=hrnet_w32|hrnet_w48|resnet50] (MVAL_CONV / MVAL_P2=force select the plan)"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.engine import _plan_for
from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet, PoseResNet, hrnet_w48

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
arch = sys.argv[2] if len(sys.argv) > 2 else "hrnet_w32"
dev = torch.device("cuda:0")
m = PoseHighResolutionNet(19) if arch == "hrnet_w32" else PoseResNet(19) if arch == "resnet50" else PoseHighResolutionNet(19, hrnet_cfg=hrnet_w48())
sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(m._graph.param_shapes(), 0).items()}
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
x = torch.randn(n, 3, 256, 256, device=dev) if arch == "hrnet_w32" else torch.randn(n, 3, 256, 192, device=dev) if arch == "resnet50" else torch.randn(n, 3, 384, 288, device=dev)
with torch.no_grad():
    m(x)
    plan = _plan_for(m, x)
    acc = None
    for _ in range(5):
        _, ms, fl = plan.forward_timed(x)
        acc = ms if acc is None else acc + ms
ms = acc / 5
g = defaultdict(lambda: [0, 0.0, 0.0])
KIND = {0: "conv", 1: "maxpool", 2: "deconv", 3: "block", 4: "to_p2", 5: "bneck", 6: "stem_p2", 7: "fuse_up"}
for o, t, f in zip(plan.ops, ms, fl):
    key = (KIND[o.kind], o.algo, o.k, o.stride, o.cin, o.cout, o.hin, o.win, o.up, int(o.res1_off >= 0) + int(o.res2_off >= 0))
    g[key][0] += 1
    g[key][1] += t
    g[key][2] += f
print(f"{arch} n={n} p2={plan.p2} forward (sum of launches) {ms.sum():.3f} ms, {len(plan.ops)} launches")
for key, (cnt, t, f) in sorted(g.items(), key=lambda kv: -kv[1][1]):
    kind, algo, k, s, ci, co, h, w, up, nres = key
    print(f"{kind:6s} algo{algo} k{k}s{s} {ci:4d}->{co:<4d} {h:3d}x{w:<3d} up{up} res{nres}  x{cnt:3d}  {t:7.3f} ms  ({t / cnt * 1e3:7.1f} us each, {f / t / 1e9 if t else 0:6.1f} TFLOP/s)")
