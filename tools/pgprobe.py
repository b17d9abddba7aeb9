#!/usr/bin/env python3
"""Does an initialised process group change the speed of the captured step?  (one GPU, single rank)
    python tools/pgprobe.py none|gloo|nccl_lazy|nccl_eager"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.distributed as dist

mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29514")
torch.cuda.set_device(0)
if mode == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
elif mode in ("nccl_lazy", "nccl_late"):
    dist.init_process_group("nccl", rank=0, world_size=1)
elif mode == "nccl_eager":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif mode == "nccl_used":
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()

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
if mode == "nccl_late":                                  # communicator created only after the graph capture
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tr.step(x, y)
torch.cuda.synchronize()
print("%-12s %.3f ms/step" % (mode, (time.perf_counter() - t0) / 30 * 1e3))
if mode != "none":
    dist.destroy_process_group()
