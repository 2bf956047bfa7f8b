"""Per-parameter gradient error of one training step against the torch-CPU oracle (which layers drift, by how
much): the diagnostic that located the coherent bias of the bf16 MFMA accumulate.  usage: grad_diag.py [arch]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, cases
from oracle import models
import test_gpu_train as T
dev = torch.device("cuda:0")
import json
c = json.loads(sys.argv[1]) if len(sys.argv) > 1 else dict(arch="hrnet_w32", seed=5, n=3, h=64, w=64, j=7)
m, _, hm, loss, sd = T._train_once(c, dev)
x, gt, valid = cases.train_input(c)
def cpu(dt):
    sdc = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k, v in sdc.items():
        if v.dtype.is_floating_point and "running" not in k: v.requires_grad_(True)
    h = (models.pose_resnet_forward(sdc, torch.from_numpy(x).to(dt), training=True) if c["arch"] == "resnet50" else models.hrnet_forward(sdc, torch.from_numpy(x).to(dt), models.HRNET_W48 if c["arch"] == "hrnet_w48" else models.HRNET_W32, training=True))
    l = models.pose_2d_mse(h, torch.from_numpy(gt).to(dt), torch.from_numpy(valid).reshape(h.shape[0], -1, 1, 1)); l.backward()
    return sdc, h.detach(), l.item()
(s64, h64, l64), (s32, h32, l32) = cpu(torch.float64), cpu(torch.float32)
hg = hm.detach().cpu().double()
print('heat-map max err vs fp64: gpu %.3e cpu32 %.3e (range %.3e); loss rel err gpu %.3e cpu32 %.3e' % ((hg - h64).abs().max().item(), (h32.double() - h64).abs().max().item(), h64.abs().max().item(), abs(loss.item() - l64) / abs(l64), abs(l32 - l64) / abs(l64)))
rows = []
for k, p in m.named_parameters():
    t = s64[k].grad.numpy()
    rows.append((T._rel(p.grad.cpu().numpy(), t) / (T._rel(s32[k].grad.numpy(), t) + 1e-4), T._rel(p.grad.cpu().numpy(), t), T._rel(s32[k].grad.numpy(), t), k))
rows.sort(reverse=True)
for r in rows[:25]: print("%.1f %.2e %.2e %s" % r)
print("median gpu", np.median([r[1] for r in rows]), "median cpu", np.median([r[2] for r in rows]))
