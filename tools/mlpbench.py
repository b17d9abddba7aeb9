#!/usr/bin/env python3
"""Per-stack timing of the fused MLP (fwd and fwd+bwd) at the benchmark shapes, against the
HBM / fp32-MFMA bounds of each stack.  GPU box only."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from khairil_tum_facade_semantic_segmentation_amd import mlp

B = 16
STACKS = [  # name, M, K1, K2, widths, pool_k, input needs grad
    ("sa1", B * 1024 * 32, 12, 0, (32, 32, 64), 32, False),
    ("sa2", B * 256 * 32, 68, 0, (64, 64, 128), 32, True),
    ("sa3", B * 64 * 32, 132, 0, (128, 128, 256), 32, True),
    ("sa4", B * 16 * 32, 260, 0, (256, 256, 512), 32, True),
    ("fp4", B * 64, 256, 512, (256, 256), 0, True),
    ("fp3", B * 256, 128, 256, (256, 256), 0, True),
    ("fp2", B * 1024, 64, 256, (256, 128), 0, True),
    ("fp1", B * 4096, 128, 0, (128, 128, 128), 0, True),
]


def graph_time(fn, reps=10):
    for _ in range(2):
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
    return a.elapsed_time(b) / reps * 1e3


tot_f = tot_fb = 0.0
for name, M, K1, K2, widths, pool, ng in STACKS:
    convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
    last = K1 + K2
    for co in widths:
        convs.append(torch.nn.Conv1d(last, co, 1))
        bns.append(torch.nn.BatchNorm1d(co))
        last = co
    convs, bns = convs.cuda().train(), bns.cuda().train()
    x1 = torch.randn(M, K1, device="cuda", requires_grad=ng)
    x2 = torch.randn(M, K2, device="cuda", requires_grad=ng) if K2 else None
    rows_out = M // pool if pool else M
    go = torch.randn(rows_out, widths[-1], device="cuda")

    def fwd():
        with torch.no_grad():
            return mlp.mlp_stack(x1, x2, convs, bns, pool)

    def fwdbwd():
        y = mlp.mlp_stack(x1, x2, convs, bns, pool)
        params = [p for m in list(convs) + list(bns) for p in m.parameters()]
        ins = [t for t in (x1, x2) if t is not None and t.requires_grad]
        torch.autograd.grad(y, params + ins, go)

    tf = graph_time(fwd)
    tfb = graph_time(fwdbwd)
    dims = [K1 + K2] + list(widths)
    flops = sum(2.0 * M * a * b for a, b in zip(dims[:-1], dims[1:]))
    zbytes = sum(4.0 * M * c for c in widths)
    inb = 4.0 * M * (K1 + K2)
    mem_f = (inb + 2 * zbytes) / 3.6e12 * 1e6          # write Z, read it once more
    mem_fb = mem_f + (inb + 5 * zbytes) / 3.6e12 * 1e6
    cmp_f = flops / 100e12 * 1e6
    print("%-4s M=%7d %-22s fwd %7.1f us (mem %6.1f, mfma@100TF %6.1f) | fwd+bwd %7.1f us (mem %6.1f, mfma %6.1f)"
          % (name, M, "x".join(map(str, dims)), tf, mem_f, cmp_f, tfb, mem_fb, 3 * cmp_f))
    tot_f += tf
    tot_fb += tfb
print("total fwd %.2f ms, fwd+bwd %.2f ms" % (tot_f / 1e3, tot_fb / 1e3))
