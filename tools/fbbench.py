#!/usr/bin/env python3
"""Timing of pn2_mlp_bwd_layer (one-pass layer backward) at the benchmark's layer shapes, against the
bytes it has to move.  GPU box only.   python tools/fbbench.py"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from khairil_tum_facade_semantic_segmentation_amd import _lib
from khairil_tum_facade_semantic_segmentation_amd.ops import _ptr, _stream

B = 16
SHAPES = [  # name, M, N, K, pooled, masked(dX + stats), dx
    ("sa1.l3", B * 1024 * 32, 64, 32, True, True, True),
    ("sa1.l2", B * 1024 * 32, 32, 32, False, True, True),
    ("sa1.l1", B * 1024 * 32, 32, 12, False, False, False),
    ("sa2.l3", B * 256 * 32, 128, 64, True, True, True),
    ("sa2.l2", B * 256 * 32, 64, 64, False, True, True),
    ("sa2.l1", B * 256 * 32, 64, 68, False, False, True),
    ("fp1.l2", B * 4096, 128, 128, False, True, True),
    ("sa3.l2", B * 64 * 32, 128, 128, False, True, True),
]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    f32 = dict(dtype=torch.float32, device=dev)
    for name, M, N, K, pooled, masked, dx in SHAPES:
        g = torch.randn((M // 32 if pooled else M, N), **f32)
        argk = torch.randint(0, 32, (M // 32, N), dtype=torch.uint8, device=dev) if pooled else None
        z = torch.randn((M, N), **f32)
        x = torch.randn((M, K), **f32)
        w = torch.randn((N, K), **f32)
        cs = [torch.rand(N, **f32) + 0.5 for _ in range(6)]
        below = [torch.rand(K, **f32) + 0.5 for _ in range(4)] if masked else [None] * 4
        P = lib.pn2_mlp_bwd_layer_partials(M, N, K)
        gp = torch.empty((M, K), **f32) if dx else None
        spart = torch.empty((P, 2, K), **f32) if masked else None
        wpart = torch.empty((P, N, K + 1), **f32)
        dw, db = torch.empty((N, K), **f32), torch.empty(N, **f32)

        def run():
            rc = lib.pn2_mlp_bwd_layer(_ptr(g), g.stride(0), _ptr(z), z.stride(0), _ptr(argk), 32 if pooled else 0,
                                       *[_ptr(c) for c in cs], _ptr(w), w.stride(0), _ptr(x), x.stride(0),
                                       *[_ptr(b) for b in below], _ptr(gp), 0 if gp is None else gp.stride(0), _ptr(spart),
                                       _ptr(wpart), _ptr(dw), _ptr(db), None, None, None, None, M, N, K, _stream(dev))
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        reps = 20
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        byts = 4 * (g.numel() + z.numel() + x.numel() + (gp.numel() if dx else 0))
        flops = 2.0 * M * N * K * (2 if dx else 1)
        print("%-7s M=%7d N=%3d K=%3d P=%4d  %7.1f us  (bytes %6.1f MB -> %5.2f TB/s; %5.1f TFLOP/s)" %
              (name, M, N, K, P, us, byts / 1e6, byts / us / 1e6, flops / us / 1e6))


if __name__ == "__main__":
    main()
