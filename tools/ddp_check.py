#!/usr/bin/env python3
"""One training step with and without DistributedDataParallel (world size 1, RCCL backend): same loss and
gradients; buffers broadcast path exercised.  Launch: python -m torch.distributed.run --nproc-per-node 1 tools/ddp_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import bench
from multi_view_active_learning_amd import synth
from multi_view_active_learning_amd.pose_estimators import Pose2DMeanSquaredError

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"])); torch.cuda.set_device(dev)
x = torch.from_numpy(synth.images(5, 2, 4, 128, 128)).reshape(8, 3, 128, 128).to(dev)
gt = torch.rand(8, 19, 32, 32, device=dev); pv = torch.ones(8, 19, 1, 1, dtype=torch.uint8, device=dev)
res = []
for use_ddp in (False, True):
    model, _ = bench.build_model("hrnet_w32", 19, dev, seed=2)
    model.train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], broadcast_buffers=True) if use_ddp else model
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for _ in range(2):
        opt.zero_grad()
        loss = Pose2DMeanSquaredError().pose_2d_mse(net(x), gt, pv)
        loss.backward()
        opt.step()
    res.append((loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()))
print("loss", res[0][0], res[1][0])
d = (res[0][1] - res[1][1]).abs().max().item()
print("max grad diff ddp vs plain:", d, "grad norm", res[0][1].norm().item())
assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and d <= 1e-6 * res[0][1].abs().max().item() + 1e-12
print("DDP OK")
dist.destroy_process_group()
