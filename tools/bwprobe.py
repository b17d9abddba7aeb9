#!/usr/bin/env python3
"""HBM bandwidth probes with plain torch kernels (fill = write only, copy = read + write, sum = read only)
at a few sizes: what a streaming kernel can reach on this GPU.  GPU box only."""
import torch

def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for mb in (25, 100, 400, 1600):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda")
    y = torch.empty(n, device="cuda")
    tf = t(lambda: x.fill_(1.0))
    tc = t(lambda: y.copy_(x))
    ts = t(lambda: x.sum())
    print("%5d MB: fill %7.1f us (%5.2f TB/s write) | copy %7.1f us (%5.2f TB/s r+w) | sum %7.1f us (%5.2f TB/s read)" %
          (mb, tf, mb * 1.048576 / tf, tc, 2 * mb * 1.048576 / tc, ts, mb * 1.048576 / ts))
