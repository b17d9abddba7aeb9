#!/usr/bin/env python3
"""Where the data-parallel step spends its extra time on ONE GPU (single-rank process group, exchange path forced):
single graph | forward/backward graph + eager all-reduce + optimizer graph | the same without the collective.
    RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 python tools/dpbench.py"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.distributed as dist

from khairil_tum_facade_semantic_segmentation_amd import synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer


def run(tag, steps=30):
    blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 18)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).cuda()
    y = torch.from_numpy(labels).cuda()
    model = M.get_model(18, 3).cuda()
    tr = SemSegTrainer(model, class_weight=torch.ones(18, device="cuda"), graphs=True, prefetch_geometry=True)
    for _ in range(6):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (tag, (time.perf_counter() - t0) / steps * 1e3))


if __name__ == "__main__":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["PN2_FORCE_DP_PATH"] = "0"
    run("single graph")
    os.environ["PN2_FORCE_DP_PATH"] = "1"
    run("two graphs + eager all-reduce")
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: None
    run("two graphs, collective skipped")
    dist.all_reduce = real
