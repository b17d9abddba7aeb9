#!/usr/bin/env python3
"""us per pn2_farthest_point_sample call at the four levels of pointnet2_sem_seg (16 blocks; GPU box only)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import ops, synth

rs = np.random.RandomState(0)
for n, s in ((4096, 1024), (1024, 256), (256, 64), (64, 16)):
    xyz = torch.from_numpy(synth.make_xyz(rs, 16, n, "cube")).cuda()
    start = torch.zeros(16, dtype=torch.int64, device="cuda")
    fn = lambda: ops.farthest_point_sample_with_xyz(xyz, s, start)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10 * 1e3)
    print("N %5d -> %4d: %7.1f us per call = %.3f us per iteration" % (n, s, float(np.median(ts)), float(np.median(ts)) / s), flush=True)
