#!/usr/bin/env python3
"""End-to-end rate of train.train_epoch on a synthetic scene: device block sampler + rotate-z + captured step with
geometry prefetch + on-device metrics, the loop of localfunctions.py:184-227.  GPU box only.
    python tools/epochbench.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, scene, synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer, train_epoch


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    _lib.load()
    rs = np.random.RandomState(0)
    P, K = 2_000_000, 13
    xyz = rs.uniform(0, 1, size=(P, 3)) * np.array([40.0, 25.0, 12.0])
    labels = rs.randint(0, K, size=P)
    rgb = [rs.randint(0, 256, size=P).astype(np.float64) for _ in range(3)]
    sampler = scene.DeviceBlockSampler(xyz, labels, rgb, ["red", "blue", "green"])
    # three more rooms of different sizes: a batch then mixes rooms by point share (train.batch_plan) -- up to four sampler
    # launches and one concatenation per step
    rooms = [sampler]
    for i, n in enumerate((600_000, 300_000, 150_000)):
        rr = np.random.RandomState(10 + i)
        rooms.append(scene.DeviceBlockSampler(rr.uniform(0, 1, size=(n, 3)) * np.array([10.0, 6.0, 9.0]), rr.randint(0, K, size=n),
                                              [rr.randint(0, 256, size=n).astype(np.float64) for _ in range(3)], ["red", "blue", "green"]))
    model = M.get_model(K, 3)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev)
    tr = SemSegTrainer(model, class_weight=torch.ones(K, device=dev), graphs=True, prefetch_geometry=True, augment=True,
                       metrics=True)
    train_epoch(tr, [sampler], 0, 12, 16)                     # eager warm-up + capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = train_epoch(tr, [sampler], 1, steps, 16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("train_epoch, one room: %.3f ms per step of 16 x 4096 points (%.1f M points/s, %.0f blocks/s) loss %.4f accuracy %.3f"
          % (dt * 1e3, 16 * 4096 / dt / 1e6, 16 / dt, out["loss"], out.get("accuracy", float("nan"))))
    train_epoch(tr, rooms, 2, 4, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = train_epoch(tr, rooms, 3, steps, 16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("train_epoch, four rooms mixed per batch: %.3f ms per step (%.1f M points/s, %.0f blocks/s) loss %.4f"
          % (dt * 1e3, 16 * 4096 / dt / 1e6, 16 / dt, out["loss"]))
    # the sampler alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        sampler.sample(16, seed=i)
    torch.cuda.synchronize()
    print("sampler alone: %.3f ms per batch of 16" % ((time.perf_counter() - t0) / 50 * 1e3))


if __name__ == "__main__":
    main()
