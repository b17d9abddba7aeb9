#!/usr/bin/env python3
"""Operator-level timing of query_ball_point + group (GPU box only): from (xyz, new_xyz, feats) to (idx, grouped),
everything a caller with no plan must launch.
    python tools/ballbench.py [reps]
For both synthetic distributions at the SA1 shape (B=16, N=4096, S=1024, K=32, D=9): every kernel of
pn2_ball_query_group_select, and the planned pair (pn2_ball_plan + pn2_ball_query_group_planned); then the
deeper levels through the library's choice.  us per call = back-to-back launches in one captured graph."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, ops, synth

NAMES = {0: "default", 1: "grid", 2: "scan", 3: "grid3"}


def timeit(fn, reps=50, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return float(np.median(ts))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    lib = _lib.load()
    B, N, S, K, D = 16, 4096, 1024, 32, 9
    algo = B * (N * 12 + S * 12 + N * D * 4 + S * K * 8 + S * K * (3 + D) * 4)
    for kind in ("cube", "facade"):
        blocks, _, starts, _ = synth.draw_case(synth.BENCH_SEED, B, N, 9, kind)
        pts = torch.from_numpy(blocks).cuda()
        xyz = pts[:, :, :3].contiguous()
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, S, torch.from_numpy(starts[0]).cuda())
        idx = torch.empty((B, S, K), dtype=torch.int64, device="cuda")
        grouped = torch.empty((B, S, K, 3 + D), dtype=torch.float32, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        ref = None
        for which in (1, 3, 2):
            def call(which=which, rows=True):
                rc = lib.pn2_ball_query_group_select(which, 0.1, K, xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(), B, N, S, D,
                                                     idx.data_ptr(), grouped.data_ptr() if rows else None, 0, err.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            call()
            torch.cuda.synchronize()
            if ref is None:
                ref = (idx.clone(), grouped.clone())
            else:
                assert torch.equal(idx, ref[0]) and torch.equal(grouped, ref[1]), NAMES[which]
            t = timeit(call, reps)
            ti = timeit(lambda: call(rows=False), reps)
            print("%-6s %-5s fused %6.2f us = %5.0f GB/s = %.3f of 8 TB/s | idx only %6.2f us" %
                  (kind, NAMES[which], t, algo / t / 1e3, algo / t / 1e3 / 8000, ti), flush=True)
        plan = ops.ball_plan(0.1, xyz, new_xyz, pts)

        def planned(with_plan=True):
            if with_plan:
                rc = lib.pn2_ball_plan(0.1, xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(), B, N, S, D, plan.buf.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            rc = lib.pn2_ball_query_group_planned(0.1, K, plan.buf.data_ptr(), xyz.data_ptr(), new_xyz.data_ptr(), pts.data_ptr(), B, N, S,
                                                  D, idx.data_ptr(), grouped.data_ptr(), 0, err.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        planned()
        torch.cuda.synchronize()
        assert torch.equal(idx, ref[0]) and torch.equal(grouped, ref[1])
        t, tq = timeit(planned, reps), timeit(lambda: planned(False), reps)
        print("%-6s plan+query %6.2f us = %.3f | query alone %6.2f us = %.3f" % (kind, t, algo / t / 8e6, tq, algo / tq / 8e6), flush=True)
        assert int(err.item()) == 0
    rs = np.random.RandomState(0)
    for (n, s, r, d) in ((1024, 256, 0.2, 64), (256, 64, 0.4, 128), (64, 16, 0.8, 256)):
        xyz = torch.from_numpy(synth.make_xyz(rs, 16, n, "cube")).cuda()
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, s)
        idx = torch.empty((16, s, 32), dtype=torch.int64, device="cuda")
        for which in (1, 2):
            def call(which=which):
                rc = lib.pn2_ball_query_group_select(which, r, 32, xyz.data_ptr(), new_xyz.data_ptr(), None, 16, n, s, 0, idx.data_ptr(),
                                                     None, 0, None, torch.cuda.current_stream().cuda_stream)
                assert rc in (0, -3), rc
                return rc
            if call() == 0:
                print("N=%d S=%d idx only %-5s %6.2f us" % (n, s, NAMES[which], timeit(call, reps)), flush=True)


if __name__ == "__main__":
    main()
