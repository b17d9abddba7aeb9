#!/usr/bin/env python3
"""Lab: which drop-in module's captured forward differs from its eager forward after the weights have moved?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from dropin_wiring import build, loss_fn
from khairil_tum_facade_semantic_segmentation_amd import synth, graphed
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
from oracle import pn2_oracle as orc
K, C, B, N = 8, 9, 2, 2048
blocks, labels, starts, cw = synth.draw_case(302, B, N, C, "facade", K)
model = build(U, K, C - 6)
filled = synth.fill_state_dict(orc.state_shapes(K, C - 6))
model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
model = model.cuda().train(); model.drop1.p = 0.0
x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).cuda()
y = torch.from_numpy(labels).cuda().view(-1)
state = {k: v.clone() for k, v in model.state_dict().items()}
taps = {}
for name, m in model.named_children():
    m.register_forward_hook(lambda mod, inp, out, name=name: taps.__setitem__(name, [t.detach().clone() for t in (out if isinstance(out, tuple) else (out,))]))


def run(enabled, steps):
    graphed.ENABLED = enabled
    model.load_state_dict(state)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    out = []
    for i in range(steps):
        opt.zero_grad()
        with U.fps_starts(starts):
            pred, _ = model(x)
        loss = loss_fn(pred.contiguous().view(-1, K), y, None)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        opt.step()
        out.append((float(loss.detach()), dict(taps), grads))
    return out
e = run(False, 5)
g = run(True, 5)
for i in range(5):
    print("step %d loss eager %.7f graphed %.7f" % (i, e[i][0], g[i][0]))
    for name in e[i][1]:
        d = max(float((a - b).abs().max()) for a, b in zip(e[i][1][name], g[i][1][name]))
        print("   out %-6s max diff %.3e" % (name, d))
    worst = sorted(((float((e[i][2][k] - g[i][2][k]).abs().max()) / (float(e[i][2][k].abs().max()) + 1e-9), k) for k in e[i][2]), reverse=True)[:4]
    print("   worst gradients (relative):", ["%s %.1e" % (k, v) for v, k in worst])
print(graphed.stats)
