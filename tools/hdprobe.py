#!/usr/bin/env python3
"""Head (conv2 + log_softmax) forward / backward timing at the benchmark shape, back-to-back launches in a graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from khairil_tum_facade_semantic_segmentation_amd import _lib, head
_lib.load()
M, K, C = 65536, 128, 18
y = torch.randn(M, K, device="cuda", requires_grad=True); w = (torch.randn(C, K, device="cuda") * 0.1).requires_grad_(True)
b = torch.zeros(C, device="cuda", requires_grad=True)
go = torch.randn(M, C, device="cuda")
def fwd(): return head.head_logits(y, w, b)
def fwdbwd():
    out = head.head_logits(y, w, b)
    torch.autograd.grad(out, (y, w, b), go)
for name, fn in (("fwd", fwd), ("fwd+bwd", fwdbwd)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); e.record(); torch.cuda.synchronize()
    print("head %s %.2f us" % (name, a.elapsed_time(e) / 20 * 1e3))
