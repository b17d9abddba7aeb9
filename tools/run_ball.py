#!/usr/bin/env python3
"""Runs the SA1 query_ball_point+group kernel a few times (target of rocprofv3 runs)."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from khairil_tum_facade_semantic_segmentation_amd import ops, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "cube"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
grouped = (sys.argv[3] != "idx") if len(sys.argv) > 3 else True
blocks, _, starts, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, kind)
pts = torch.from_numpy(blocks).cuda()
xyz = pts[:, :, :3].contiguous()
_, new_xyz = ops.farthest_point_sample_with_xyz(xyz, 1024, torch.from_numpy(starts[0]).cuda())
for _ in range(reps):
    ops._ball_query_group_raw(0.1, 32, xyz, new_xyz, pts if grouped else None, grouped)
torch.cuda.synchronize()
print("done")
