#!/usr/bin/env python3
"""What HBM delivers for pure-write / copy / read-reduce streams of the size of a 256-channel 64x64 activation
of 128 images (537 MB): the yardstick for the write-heavy 1x1 convs (64 -> 256)."""
import torch

dev = torch.device("cuda:0")
n = 128 * 64 * 64 * 256
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
small = torch.randn(n // 4, device=dev)


def t(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


b = n * 4
us = t(lambda: y.fill_(1.0))
print(f"fill  (write {b / 1e6:.0f} MB): {us:7.1f} us  {b / us / 1e3:6.0f} GB/s")
us = t(lambda: y.copy_(x))
print(f"copy  (read + write {2 * b / 1e6:.0f} MB): {us:7.1f} us  {2 * b / us / 1e3:6.0f} GB/s")
us = t(lambda: x.sum())
print(f"sum   (read {b / 1e6:.0f} MB): {us:7.1f} us  {b / us / 1e3:6.0f} GB/s")
us = t(lambda: torch.add(x, y, out=y))
print(f"add   (2 reads + write {3 * b / 1e6:.0f} MB): {us:7.1f} us  {3 * b / us / 1e3:6.0f} GB/s")
v = y.view(-1, 4, n // 4 // (n // 4) if False else 1) if False else None
us = t(lambda: torch.relu_(y))
print(f"relu_ (read + write in place {2 * b / 1e6:.0f} MB): {us:7.1f} us  {2 * b / us / 1e3:6.0f} GB/s")
