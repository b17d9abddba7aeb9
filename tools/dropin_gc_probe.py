import gc, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from dropin_wiring import build, loss_fn
from khairil_tum_facade_semantic_segmentation_amd import ops, synth, graphed
from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
dev = torch.device("cuda:0")
blocks, labels, _, _ = synth.draw_case(synth.BENCH_SEED, 16, 4096, 9, "cube", 18)
x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev); y = torch.from_numpy(labels).to(dev).view(-1)
model = build(U, 18, 3).to(dev).train()
cw = torch.ones(18, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
import time
def step():
    opt.zero_grad(); pred, _ = model(x); loss = loss_fn(pred.contiguous().view(-1, 18), y, cw); loss.backward(); opt.step(); return loss
for i in range(8):
    step()
print("gc enabled:", gc.isenabled(), "capturing:", ops.capturing(), "counts", gc.get_count(), "stats", graphed.stats)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 10, "gc", gc.get_stats()[2])
gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): step()
torch.cuda.synchronize(); print("ms/step with the collector off", (time.perf_counter() - t0) * 10)
