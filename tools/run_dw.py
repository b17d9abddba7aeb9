#!/usr/bin/env python3
"""Runs only the weight-gradient kernel (pn2_mlp_dw, slabs only) of one deep-layer shape a few times -- the target of
tools/bqlab/pmc.sh passes:  python3 tools/run_dw.py sa4.2 [reps]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from khairil_tum_facade_semantic_segmentation_amd import _lib, mlp

SHAPES = {"sa3.2": (32768, 256, 128, 32), "sa4.2": (8192, 512, 256, 32), "sa4.1": (8192, 256, 256, 0), "fp2.0": (16384, 256, 384, 0),
          "fp3.0": (4096, 256, 512, 0)}
M, Co, Ci, pool = SHAPES[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = _lib.load()
f32 = dict(dtype=torch.float32, device="cuda")
p = mlp._ptr
torch.manual_seed(0)
rows_g = M // pool if pool else M
g = torch.randn(rows_g, Co, **f32)
argk = torch.randint(0, pool, (rows_g, Co), dtype=torch.uint8, device="cuda") if pool else None
z, x = torch.randn(M, Co, **f32), torch.randn(M, Ci, **f32)
cs = [torch.rand(Co, **f32) + 0.5, torch.randn(Co, **f32) * 0.1, torch.randn(Co, **f32) * 0.1, torch.rand(Co, **f32) + 0.5,
      torch.randn(Co, **f32) * 0.01, torch.randn(Co, **f32) * 0.01]
below = [torch.rand(Ci, **f32) + 0.5, torch.randn(Ci, **f32) * 0.1]
Pw = lib.pn2_mlp_dw_partials(M, Co, Ci)
wpart = torch.empty((Pw, Co, Ci + 1), **f32)
for _ in range(reps):
    rc = lib.pn2_mlp_dw(p(g), g.stride(0), p(z), z.stride(0), p(argk), pool, p(cs[0]), p(cs[1]), p(cs[2]), p(cs[3]), p(cs[4]),
                        p(cs[5]), p(x), x.stride(0), Ci, None, 0, 0, p(below[0]), p(below[1]), M, Co, p(wpart), None, None, None)
    assert rc == 0, rc
torch.cuda.synchronize()
print("ok", sys.argv[1], "P", Pw)
