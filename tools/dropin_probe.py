#!/usr/bin/env python3
"""Where the drop-in step's time goes (GPU box): forward / backward / optimizer wall time of the reference-wired model on
the HIP pointnet2_utils, host-side enqueue time against GPU time, with the modules' own graphs on and off.
    python tools/dropin_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

from dropin_wiring import build, loss_fn
from khairil_tum_facade_semantic_segmentation_amd import synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U


def main():
    dev = torch.device("cuda:0")
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 18)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev).view(-1)
    model = build(U, 18, 3)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    cw = torch.ones(18, device=dev)
    from khairil_tum_facade_semantic_segmentation_amd import graphed
    for graphs in (True, False, True, False):
        graphed.ENABLED = graphs
        for _ in range(6):
            opt.zero_grad()
            loss_fn(model(x)[0].contiguous().view(-1, 18), y, cw).backward()
            opt.step()
        torch.cuda.synchronize()
        tf = tb = to = hf = hb = 0.0
        n = 10
        for _ in range(n):
            opt.zero_grad()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred, _ = model(x)
            loss = loss_fn(pred.contiguous().view(-1, 18), y, cw)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            loss.backward()
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            hf += t1 - t0; tf += t2 - t0; hb += t3 - t2; tb += t4 - t2; to += t5 - t4
        print("module graphs=%-5s forward %.2f ms (host enqueue %.2f) | backward %.2f ms (host %.2f) | optimizer %.2f ms | sum %.2f ms"
              % (graphs, tf / n * 1e3, hf / n * 1e3, tb / n * 1e3, hb / n * 1e3, to / n * 1e3, (tf + tb + to) / n * 1e3), flush=True)


def profile():
    import cProfile
    import pstats
    dev = torch.device("cuda:0")
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 18)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev).view(-1)
    model = build(U, 18, 3)
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    cw = torch.ones(18, device=dev)

    def step():
        opt.zero_grad()
        loss_fn(model(x)[0].contiguous().view(-1, 18), y, cw).backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "profile":
        profile()
    else:
        main()
