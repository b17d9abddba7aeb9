#!/usr/bin/env python3
"""Is the captured training step bound by the GPU or by the host's graph launch?  Times the replay loop on
the host (no synchronisation) and the same loop including the final synchronisation."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer

blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 18)
x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).cuda()
y = torch.from_numpy(labels).cuda()
model = M.get_model(18, 3).cuda()
tr = SemSegTrainer(model, class_weight=torch.ones(18, device="cuda"), graphs=True, prefetch_geometry=True)
for _ in range(6):
    tr.step(x, y)
torch.cuda.synchronize()
for steps in (10, 30, 100):
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%3d steps: host loop %.3f ms/step, with final sync %.3f ms/step" % (steps, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
g = tr._g_fwd_bwd
torch.cuda.synchronize()
t0 = time.perf_counter()
g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("one replay: launch call %.3f ms, until done %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
