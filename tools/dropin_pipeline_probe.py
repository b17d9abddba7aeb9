#!/usr/bin/env python3
"""Drop-in step WITHOUT synchronisation between its parts (the real loop): wall time per step and the host's time inside
forward / backward / optimizer, with the modules' own graphs and with eager modules.   python tools/dropin_pipeline_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

from dropin_wiring import build, loss_fn
from khairil_tum_facade_semantic_segmentation_amd import graphed, synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U

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
for name, graphs in (("modules replay their graphs", True), ("eager modules", False), ("modules replay their graphs", True), ("eager modules", False)):
    graphed.ENABLED = graphs
    for _ in range(8):
        opt.zero_grad()
        loss_fn(model(x)[0].contiguous().view(-1, 18), y, cw).backward()
        opt.step()
    torch.cuda.synchronize()
    n = 60
    hf = hb = ho = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        opt.zero_grad()
        pred, _ = model(x)
        loss = loss_fn(pred.contiguous().view(-1, 18), y, cw)
        b = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        opt.step()
        d = time.perf_counter()
        hf += b - a; hb += c - b; ho += d - c
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("%-30s wall %.3f ms/step | host %.3f ms/step (forward %.2f, backward %.2f, optimizer %.2f)" %
          (name, wall / n * 1e3, host / n * 1e3, hf / n * 1e3, hb / n * 1e3, ho / n * 1e3), flush=True)
