#!/usr/bin/env python3
"""Shader clock / power while a kernel family runs back to back: does the chip hold its clock under these kernels?
A child process samples `rocm-smi --showclocks --showpower` every ~0.2 s while this process loops one layer
(tools/conv_sweep.py's launch) for a few seconds.  usage: clock_probe.py [h2|bf3|block32|idle]"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1] if len(sys.argv) > 1 else "h2"
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            samples.append(("err", str(e)))
            return
        sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.findall(r"Power \(W\): ([\d.]+)", out) or re.findall(r"Graphics Package Power \(W\): ([\d.]+)", out)
        samples.append((sclk[0] if sclk else None, pw[0] if pw else None))
        time.sleep(0.2)


import torch  # noqa: E402

from multi_view_active_learning_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = 128
if what == "block32":
    x = torch.relu(torch.randn(n, 64, 64, 32, device=dev))
    w1 = torch.randn(32, 32, 3, 3, device=dev) * 0.08
    one, zero = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    fn = lambda: ops.fused_basic_block(x, w1, one, zero, w1, one, zero)
elif what == "idle":
    fn = lambda: time.sleep(0.01)
else:
    x = torch.randn(n, 16, 16, 128, device=dev)
    w1 = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    algo = ops.ALGO_MFMA_H2 if what == "h2" else ops.ALGO_MFMA_BF3
    fn = lambda: ops.fused_conv(x, w1, one, zero, relu=True, algo=algo)
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
k = 0
while time.time() - t0 < 4.0:
    fn()
    k += 1
torch.cuda.synchronize()
stop = True
th.join()
print(what, "launches", k, "samples (sclk MHz, W):", samples[:3], "...", samples[-8:])
