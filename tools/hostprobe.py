#!/usr/bin/env python3
"""Is the captured training step bound by the host?  (GPU box only.)
    python tools/hostprobe.py [steps]
Host time inside trainer.step() (copies into the static buffers + hipGraphLaunch, no synchronisation) against the
wall time per step with the queue kept full."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

sys.argv = [sys.argv[0]] + sys.argv[1:]
from tools.overlap_probe import build            # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    tr, x, y = build(dev, False)
    for _ in range(8):
        tr.step(x, y)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        tr.step(x, y)
        host += time.perf_counter() - a
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("per step: host inside step() %.3f ms | issue loop %.3f ms | wall %.3f ms" % (host / steps * 1e3, t_issue / steps * 1e3, wall / steps * 1e3))
    # the graph launch alone
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr._g_fwd_bwd.replay()
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    print("one replay from an idle queue: launch call %.3f ms, until done %.3f ms" % ((b - a) * 1e3, (c - a) * 1e3))


if __name__ == "__main__":
    main()
