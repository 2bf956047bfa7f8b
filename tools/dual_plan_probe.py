#!/usr/bin/env python3
"""Does a second forward in flight raise the throughput?  Two copies of the model (two plans: two arenas) on two torch streams, batches
alternating between them, against one model on one stream.  usage: dual_plan_probe.py [arch=hrnet_w32] [n=128] [h=256] [w=256] [batches=100]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

arch = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
h = int(sys.argv[3]) if len(sys.argv) > 3 else 256
w = int(sys.argv[4]) if len(sys.argv) > 4 else 256
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda:0")
models = [bench.build_model(arch, 19, dev, seed=4)[0] for _ in range(2)]
x = torch.randn(n, 3, h, w, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run(k):
    with torch.no_grad():
        for i in range(6):
            models[i % k](x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nb):
            j = i % k
            if k == 1:
                models[0](x)
            else:
                with torch.cuda.stream(streams[j]):
                    models[j](x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nb * 1e3


for rep in range(2):
    print(f"{arch} n={n} {h}x{w}: one plan {run(1):.3f} ms per batch; two plans on two streams {run(2):.3f} ms per batch", flush=True)
