#!/usr/bin/env python3
"""Bound / actual maximum of every P2 activation of a plan (csrc/conv_p2.h: the scale of a P2 tensor comes from an a-priori
bound), per tensor: log2 of the largest ratio over the images.  usage: p2_slack.py [n_images=128] [seed=0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

os.environ["MVAL_P2_SLACK_CHECK"] = "0"
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.engine import P2_ROW, _plan_for
from multi_view_active_learning_amd.pose_estimators import PoseHighResolutionNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
m = PoseHighResolutionNet(19)
sd = {k: torch.from_numpy(v) for k, v in synth.synthetic_state_dict(m._graph.param_shapes(), seed).items()}
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
x = torch.from_numpy(synth.images(1000, n // 4, 4, 256, 256)).reshape(n, 3, 256, 256).to(dev)
with torch.no_grad():
    m(x)
plan = _plan_for(m, x)
print("plan.p2", plan.p2, "overall log2 slack", plan.p2_slack_log2())
g = plan.graph
# activation id -> rows offset: rebuild as the plan did
off = plan.amax_base
rows_of = {}
dims = {}
for a in g.acts:
    pass
ar = plan.arena.view(torch.int32)
res = []
for ro in plan._p2_rows:
    rows = ar[ro : ro + n * P2_ROW].reshape(n, P2_ROW)
    amax = rows[:, : P2_ROW // 2].view(torch.float32).max(dim=1).values
    inv = rows[:, P2_ROW - 1 : P2_ROW].view(torch.float32)[:, 0]
    ok = (amax > 0) & (inv > 0)
    if not bool(ok.any()):
        continue
    sl = torch.log2(8192.0 * inv[ok] / amax[ok])
    res.append((float(sl.max()), float(sl.min()), float(amax[ok].min()), float(amax[ok].max()), ro, int(ok.sum())))
res.sort(reverse=True)
# which op writes which rows
by_rows = {}
for i, o in enumerate(plan.ops):
    by_rows.setdefault(int(o.out_amax_off), []).append((i, o.kind, o.k, o.stride, o.cin, o.cout, o.hout, o.wout, o.up))
for mx, mn, amin, amx, ro, cnt in res[:25]:
    print(f"slack max 2^{mx:5.2f} min 2^{mn:5.2f}  amax {amin:9.3e} .. {amx:9.3e}  images {cnt:3d}  writer(s) {by_rows.get(ro)}")
