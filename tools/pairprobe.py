#!/usr/bin/env python3
"""Would the dW kernel and the dX GEMM of one layer overlap if they ran side by side?  (GPU box only.)
    python tools/pairprobe.py [reps]
For the deep-level layer shapes (rows M, outputs Co, inputs Ci) whose backward runs as two launches (pn2_mlp_dw with
deferred slab sums, then pn2_mlp_gemm with the BatchNorm-backward prologue): us per layer for the two launches back to
back on one stream (a captured graph of `reps` pairs), each alone, and both streams running their `reps` launches at the
same time (eager, two streams: an upper bound of what a single combined launch could reach)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, mlp

SHAPES = [("sa3.2", 32768, 256, 128, 32), ("sa4.2", 8192, 512, 256, 32), ("sa4.1", 8192, 256, 256, 0), ("sa4.0", 8192, 256, 260, 0),
          ("fp2.0", 16384, 256, 384, 0), ("fp3.1", 4096, 256, 256, 0), ("fp3.0", 4096, 256, 512, 0), ("fp4.1", 1024, 256, 256, 0),
          ("fp4.0", 1024, 256, 768, 0)]


def graph_time(fn, reps):
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    lib = _lib.load()
    dev = torch.device("cuda:0")
    f32 = dict(dtype=torch.float32, device=dev)
    p = mlp._ptr
    for name, M, Co, Ci, pool in SHAPES:
        torch.manual_seed(0)
        rows_g = M // pool if pool else M
        g = torch.randn(rows_g, Co, **f32)
        argk = torch.randint(0, pool, (rows_g, Co), dtype=torch.uint8, device=dev) if pool else None
        z, x, zp = torch.randn(M, Co, **f32), torch.randn(M, Ci, **f32), torch.randn(M, Ci, **f32)
        w = torch.randn(Co, Ci, **f32) * 0.05
        cs = [torch.rand(Co, **f32) + 0.5, torch.randn(Co, **f32) * 0.1, torch.randn(Co, **f32) * 0.1, torch.rand(Co, **f32) + 0.5,
              torch.randn(Co, **f32) * 0.01, torch.randn(Co, **f32) * 0.01]
        below = [torch.rand(Ci, **f32) + 0.5, torch.randn(Ci, **f32) * 0.1, torch.randn(Ci, **f32) * 0.1, torch.rand(Ci, **f32) + 0.5]
        Pw = lib.pn2_mlp_dw_partials(M, Co, Ci)
        wpart = torch.empty((Pw, Co, Ci + 1), **f32)
        gp = torch.empty((M, Ci), **f32)
        P = lib.pn2_mlp_gemm_max_partials(M)
        part = torch.empty((P, 2, Ci), **f32)

        def dw(stream=None):
            s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            rc = lib.pn2_mlp_dw(p(g), g.stride(0), p(z), z.stride(0), p(argk), pool, p(cs[0]), p(cs[1]), p(cs[2]), p(cs[3]), p(cs[4]),
                                p(cs[5]), p(x), x.stride(0), Ci, None, 0, 0, p(below[0]), p(below[1]), M, Co, p(wpart), None, None, s)
            assert rc == 0, rc

        def dx(stream=None):
            s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            rc = lib.pn2_mlp_gemm(p(g), g.stride(0), Co, p(z), z.stride(0), Co, mlp.PRO_BN_BWD, p(cs[0]), p(cs[1]), p(cs[2]), p(cs[3]),
                                  p(cs[4]), p(cs[5]), p(argk), pool, p(w), w.stride(0), 1, None, p(gp), gp.stride(0), None, 0, 0, M, Ci,
                                  p(part), p(zp), zp.stride(0), p(below[0]), p(below[1]), p(below[2]), p(below[3]), s)
            assert rc == 0, rc

        t_pair = graph_time(lambda: (dw(), dx()), reps)
        t_dw, t_dx = graph_time(dw, reps), graph_time(dx, reps)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            s1.wait_stream(torch.cuda.current_stream())
            s2.wait_stream(torch.cuda.current_stream())
            for _ in range(reps):
                dw(s1.cuda_stream)
                dx(s2.cuda_stream)
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.current_stream().wait_stream(s2)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps * 1e3)
        print("%-6s M %6d Co %4d Ci %4d | dW %6.1f  dX %6.1f  back to back %6.1f | two streams at once %6.1f us per pair" %
              (name, M, Co, Ci, t_dw, t_dx, t_pair, best), flush=True)


if __name__ == "__main__":
    main()
