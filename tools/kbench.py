#!/usr/bin/env python3
"""Developer micro-benchmark of the individual HIP kernels (GPU box only):
    python tools/kbench.py [ball|fps|nn|group|all]
Times back-to-back launches with events on the launch stream; prints us/launch and GB/s."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, ops, synth


def timeit(fn, reps=20, warm=3):
    """us per call, GPU time only: the calls are captured into one hipGraph (the launchers are
    capture-safe) so host launch overhead does not pollute short kernels."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def bench_ball():
    lib = _lib.load()
    for kind in ("cube", "facade"):
        blocks, _, starts, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, kind)
        pts = torch.from_numpy(blocks).cuda()
        xyz = pts[:, :, :3].contiguous()
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, 1024, torch.from_numpy(starts[0]).cuda())
        algo = 16 * (4096 * 12 + 1024 * 12 + 4096 * 36 + 1024 * 32 * 8 + 1024 * 32 * 48)
        t_full = timeit(lambda: ops._ball_query_group_raw(0.1, 32, xyz, new_xyz, pts, True))
        t_idx = timeit(lambda: ops._ball_query_group_raw(0.1, 32, xyz, new_xyz, None, False))
        idx, _ = ops._ball_query_group_raw(0.1, 32, xyz, new_xyz, None, False)
        t_grp = timeit(lambda: ops.group_points(xyz, new_xyz, pts, idx))
        print("ball %-6s SA1 B16: fused %.1f us (%.0f GB/s, %.1f%% of 8TB/s) | idx only %.1f us | separate group kernel %.1f us"
              % (kind, t_full, algo / t_full / 1e3, algo / t_full / 1e3 / 80, t_idx, t_grp))
    # deeper levels (cube)
    rs = np.random.RandomState(0)
    for (N, S, r, D) in ((1024, 256, 0.2, 64), (256, 64, 0.4, 128), (64, 16, 0.8, 256)):
        xyz = torch.from_numpy(synth.make_xyz(rs, 16, N, "cube")).cuda()
        pts = torch.randn(16, N, D, device="cuda")
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, S)
        t_full = timeit(lambda: ops._ball_query_group_raw(r, 32, xyz, new_xyz, pts, True))
        algo = 16 * (N * 12 + S * 12 + N * D * 4 + S * 32 * 8 + S * 32 * (3 + D) * 4)
        print("ball N=%d S=%d D=%d: %.1f us (%.0f GB/s)" % (N, S, D, t_full, algo / t_full / 1e3))


def bench_fps():
    rs = np.random.RandomState(0)
    for (N, S) in ((4096, 1024), (1024, 256), (256, 64), (64, 16)):
        for B in (16, 64):
            xyz = torch.from_numpy(synth.make_xyz(rs, B, N, "cube")).cuda()
            st = torch.zeros(B, dtype=torch.long, device="cuda")
            t = timeit(lambda: ops.farthest_point_sample_with_xyz(xyz, S, st), reps=10, warm=2)
            print("fps B=%d N=%d S=%d: %.1f us (%.3f us/iter)" % (B, N, S, t, t / S))


def bench_nn():
    rs = np.random.RandomState(0)
    for (N, S, D) in ((4096, 1024, 128), (1024, 256, 256), (256, 64, 256), (64, 16, 512)):
        x1 = torch.from_numpy(synth.make_xyz(rs, 16, N, "cube")).cuda()
        x2 = x1[:, :S].contiguous()
        p2 = torch.randn(16, S, D, device="cuda")
        t = timeit(lambda: ops.three_nn(x1, x2))
        idx3, w3 = ops.three_nn(x1, x2)
        t2 = timeit(lambda: ops.three_interpolate(p2, idx3, w3))
        g = torch.randn(16, N, D, device="cuda")
        p2.requires_grad_(True)
        # forward AND backward inside the captured region.  (Round 1 built `out` outside and differentiated it inside
        # the capture: the autograd engine then runs the backward on the FORWARD's stream -- the legacy default
        # stream -- from its worker thread while another stream is in global-mode capture, which HIP forbids
        # (hipErrorStreamCaptureImplicit, capture invalidated; with an allocator miss also a hipMalloc during capture).
        # That is what dumped core in profiles/r01/kbench_v0.log)
        t3 = timeit(lambda: torch.autograd.grad(ops.three_interpolate(p2, idx3, w3), p2, g), reps=5) - t2
        p2.requires_grad_(False)
        print("three_nn N=%d S=%d: %.1f us | interpolate D=%d: %.1f us | backward %.1f us" % (N, S, t, D, t2, t3))


def bench_group():
    """The separate grouping pass the model's main stream runs at every level (indices come from the geometry
    stream), at the padded pitches the MLP reads."""
    rs = np.random.RandomState(0)
    for (N, S, r, D) in ((4096, 1024, 0.1, 9), (1024, 256, 0.2, 64), (256, 64, 0.4, 128), (64, 16, 0.8, 256)):
        xyz = torch.from_numpy(synth.make_xyz(rs, 16, N, "cube")).cuda()
        pts = torch.randn(16, N, D, device="cuda")
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, S)
        idx, _ = ops._ball_query_group_raw(r, 32, xyz, new_xyz, None, False)
        ldg = (3 + D + 3) // 4 * 4
        t = timeit(lambda: ops.group_points(xyz, new_xyz, pts, idx, pad_to=4))
        out_bytes = 16 * S * 32 * ldg * 4
        print("group N=%d S=%d D=%d pitch %d: %.1f us (%.1f MB written, %.0f GB/s)" % (N, S, D, ldg, t, out_bytes / 1e6, out_bytes / t / 1e3))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("ball", "all"):
        bench_ball()
    if what in ("fps", "all"):
        bench_fps()
    if what in ("nn", "all"):
        bench_nn()
    if what in ("group", "all"):
        bench_group()
