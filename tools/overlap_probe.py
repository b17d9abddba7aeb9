#!/usr/bin/env python3
"""What does the geometry branch cost the training step?  (GPU box only.)
    python tools/overlap_probe.py [steps]
Times the captured step of bench.py's workload three ways:
  both    -- as shipped: the next batch's FPS / ball-query / 3-NN pyramid on a parallel branch of the graph;
  main    -- the same graph with the side branch issuing NO kernels (it hands the current pyramid back): the time of the
             forward + backward + Adam branch alone;
  geo     -- the pyramid alone, captured in a graph of its own.
both - main is what the concurrency costs the critical branch (dispatch arbitration + contention for CUs / HBM)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer


def build(dev, empty_side):
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 13)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev)
    model = M.get_model(13, 3)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev)
    tr = SemSegTrainer(model, class_weight=torch.ones(13, device=dev), graphs=True, prefetch_geometry=True)
    if empty_side:
        real = tr._geometry_of
        state = {"n": 0, "first": None}

        def fake(blocks_cf):
            # call 1 of _capture computes the static pyramid; the in-graph call gets it back without a launch
            state["n"] += 1
            if tr._g_fwd_bwd is None and torch.cuda.is_current_stream_capturing():
                return state["first"]
            geo = real(blocks_cf)
            state["first"] = geo
            return geo
        tr._geometry_of = fake
    return tr, x, y


def time_steps(tr, x, y, steps):
    for _ in range(6):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    _lib.load()
    tr, x, y = build(dev, False)
    both = time_steps(tr, x, y, steps)
    geo_graph = torch.cuda.CUDAGraph()
    with torch.no_grad():
        tr.model.compute_geometry(x)
        torch.cuda.synchronize()
        with torch.cuda.graph(geo_graph):
            tr.model.compute_geometry(x)
    geo_graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        geo_graph.replay()
    torch.cuda.synchronize()
    geo = (time.perf_counter() - t0) / steps * 1e3
    del tr
    tr2, x, y = build(dev, True)
    main_only = time_steps(tr2, x, y, steps)
    print("step with the geometry branch %.3f ms | forward+backward+Adam branch alone %.3f ms | pyramid alone %.3f ms"
          % (both, main_only, geo))
    print("the parallel branch costs the critical one %.3f ms" % (both - main_only))


if __name__ == "__main__":
    main()
