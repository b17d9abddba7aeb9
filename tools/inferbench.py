#!/usr/bin/env python3
"""Whole-scene inference rate (blocks/s) of the eval forward on synthetic 4096-point blocks: eager launches against the
replayed graph of scene.BlockInferencer (next sub-batch's pyramid on a parallel branch).  GPU box only.
    python tools/inferbench.py [batches]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, scene, synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    _lib.load()
    model = M.get_model(13, 3)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev).eval()
    for B in (16, 32, 64):
        blocks, _, _, _ = synth.draw_case(synth.BENCH_SEED, B, 4096, 9, "cube", 13)
        x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
        votes = scene.VotePool(B * 4096, 13, dev)
        pidx = torch.arange(B * 4096, device=dev).reshape(B, 4096)
        w = torch.ones((B, 4096), device=dev)
        with torch.no_grad():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nb):
                logp, _ = model(x)
                votes.add(logp=logp, point_idx=pidx, weight=w)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / nb
        engine = scene.BlockInferencer(model, B, 9, 4096)
        batches = [x] * nb
        engine.run(batches[:2], lambda i, logp: None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.run(batches, lambda i, logp: votes.add(logp=logp, point_idx=pidx, weight=w))
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / nb
        print("sub-batch %2d x 4096: eager %.2f ms = %6.0f blocks/s | graph + prefetch %.2f ms = %6.0f blocks/s (%.1f M points/s)"
              % (B, eager * 1e3, B / eager, graph * 1e3, B / graph, B * 4096 / graph / 1e6))


if __name__ == "__main__":
    main()
