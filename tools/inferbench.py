#!/usr/bin/env python3
"""Whole-scene inference rate (blocks/s) of the eval forward on synthetic 4096-point blocks: eager launches against the
replayed graph of scene.BlockInferencer (next sub-batch's pyramid on a parallel branch).  GPU box only.
    python tools/inferbench.py [batches]
    python tools/inferbench.py --end-to-end [points]     a raw scene -> device tiler -> network -> votes -> labels"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, scene, synth
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M


def end_to_end(P):
    """tile + infer + vote from a RAW scene (nothing pre-tiled): scene.DeviceSceneTiler -> infer_scene(graphs=True).  The
    host tiler (scene.SceneTiler, the reference's loop with a grid index) is timed beside it on a tenth of the scene."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_scene import make_scene
    dev = torch.device("cuda:0")
    K = 8
    side = float(np.sqrt(P / 16000.0))                      # ~16 k points per square metre, like the epoch bench's scene
    xyz, labels, rgb = make_scene(7, P, extent=(side, side, 3.0), num_classes=K)
    names = ["red", "blue", "green"]
    model = M.get_model(K, 3)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev).eval()
    t0 = time.perf_counter()
    tiler = scene.DeviceSceneTiler(xyz, labels, rgb, names)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    for B in (32, 64):
        data, lab, wt, idx = tiler.tile(seed=1)
        engine = scene.BlockInferencer(model, B, data.shape[2], data.shape[1])      # one captured graph per model, reused
        scene.infer_scene(model, data[:2 * B], idx[:2 * B], wt[:2 * B], P, K, batch_size=B, engine=engine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data, lab, wt, idx = tiler.tile(seed=2)
        torch.cuda.synchronize()
        t_tile = time.perf_counter() - t0
        nb = data.shape[0]
        t0 = time.perf_counter()
        pred = scene.infer_scene(model, data, idx, wt, P, K, batch_size=B, engine=engine)
        torch.cuda.synchronize()
        t_inf = time.perf_counter() - t0
        m = scene.scene_metrics(pred, labels, K)
        print("end-to-end, %d points, %d blocks, sub-batch %d: tile %.1f ms (%.0f blocks/s) + infer+vote %.1f ms -> %.0f blocks/s "
              "(%.1f M points/s); scene upload + grid index %.0f ms once; mIoU %.4f"
              % (P, nb, B, t_tile * 1e3, nb / t_tile, t_inf * 1e3, nb / (t_tile + t_inf), nb * 4096 / (t_tile + t_inf) / 1e6, t_build * 1e3,
                 m["mIoU"]), flush=True)
    sub = max(P // 10, 20000)
    hx, hl, hr = make_scene(7, sub, extent=(side / np.sqrt(10.0), side / np.sqrt(10.0), 3.0), num_classes=K)
    t0 = time.perf_counter()
    hd = scene.SceneTiler(hx, hl, hr, names).tile()[0]
    th = time.perf_counter() - t0
    print("host tiler (the reference's loop + grid index), %d points: %d blocks in %.2f s = %.0f blocks/s" % (sub, hd.shape[0], th, hd.shape[0] / th))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--end-to-end":
        _lib.load()
        return end_to_end(int(sys.argv[2]) if len(sys.argv) > 2 else 2000000)
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
