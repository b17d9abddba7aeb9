#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE on CPU (build container only).

Imports /root/reference/models/pointnet2_utils.py and pointnet2_sem_seg.py, feeds them inputs
drawn from numpy RandomState seeds (regenerable anywhere from the seed via the package's
synth.py), and stores only the reference's OUTPUTS.  The reference draws FPS start indices with
torch.randint inside farthest_point_sample (models/pointnet2_utils.py:75); this script
substitutes a queue of explicit start vectors for that call so the starts are part of the
fixture.  Nothing here travels to the GPU box except the .npz files it writes.

    python oracle/make_golden.py            # writes tests/golden/
"""
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.append(os.path.join(REF, "models"))

import torch  # noqa: E402

import khairil_tum_facade_semantic_segmentation_amd as pn2  # noqa: E402
from khairil_tum_facade_semantic_segmentation_amd import synth  # noqa: E402
from oracle import pn2_oracle as orc  # noqa: E402

U = importlib.import_module("models.pointnet2_utils")
M = importlib.import_module("pointnet2_sem_seg")
OUT = os.path.join(REPO, "tests", "golden")


class injected_fps_starts:
    """Replace torch.randint for the duration of a reference call by a FIFO of start vectors."""

    def __init__(self, starts):
        self.queue = [torch.as_tensor(np.asarray(s), dtype=torch.long) for s in starts]

    def __enter__(self):
        self._orig = torch.randint

        def fake(low, high, size, dtype=torch.long, **kw):
            s = self.queue.pop(0)
            assert tuple(s.shape) == tuple(size) and int(s.max()) < high
            return s.clone()

        torch.randint = fake
        return self

    def __exit__(self, *a):
        torch.randint = self._orig
        assert not self.queue, "unused FPS starts"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-34s %8.1f KB" % (name, os.path.getsize(path) / 1024))


LEVELS = ((1024, 0.1), (256, 0.2), (64, 0.4), (16, 0.8))
NSAMPLE = 32


def gen_geometry(kind, seed):
    """FPS chain, ball query at the four SA levels, grouping, 3-NN at the four FP shapes."""
    B, N = 2, 4096
    blocks, _, starts, _ = synth.draw_case(seed, B, N, 9, kind)
    xyz = blocks[:, :, :3].copy()
    out = {"seed": np.int64(seed), "B": np.int64(B), "N": np.int64(N)}
    xyz_lv = [xyz]
    cur = t(xyz)
    for lv, ((npoint, radius), st) in enumerate(zip(LEVELS, starts), start=1):
        with injected_fps_starts([st]):
            fps = U.farthest_point_sample(cur, npoint)
        new_xyz = U.index_points(cur, fps)
        idx = U.query_ball_point(radius, NSAMPLE, cur, new_xyz)
        out["start%d" % lv] = st
        out["fps%d" % lv] = fps.numpy().astype(np.int16)
        out["ball%d" % lv] = idx.numpy().astype(np.int16)
        # cross-check the oracle while we are here
        assert np.array_equal(orc.farthest_point_sample(cur.numpy(), npoint, st), fps.numpy()), (kind, lv, "fps")
        assert np.array_equal(orc.query_ball_point(radius, NSAMPLE, cur.numpy(), new_xyz.numpy()), idx.numpy()), (kind, lv, "ball")
        if lv == 1:
            with injected_fps_starts([st]):
                nx, grouped = U.sample_and_group(npoint, radius, NSAMPLE, cur, t(blocks))
            g = grouped.numpy()
            out["group1_rows"] = g[:, ::41].copy()                      # every 41st centroid, all K, all C
            out["group1_sum"] = np.float64(g.astype(np.float64).sum())
            out["group1_abs_sum"] = np.float64(np.abs(g.astype(np.float64)).sum())
            assert np.array_equal(orc.group_points(xyz, nx.numpy(), blocks, idx.numpy()), g)
        cur = new_xyz
        xyz_lv.append(new_xyz.numpy())
    # three-NN at FP4..FP1: xyz1 = level lv, xyz2 = level lv+1 (pointnet2_sem_seg.py:31-34)
    frs = np.random.RandomState(seed + 1)
    for lv in (3, 2, 1, 0):
        x1, x2 = t(xyz_lv[lv]), t(xyz_lv[lv + 1])
        d = U.square_distance(x1, x2)
        ds, di = d.sort(dim=-1)
        ds, di = ds[:, :, :3], di[:, :, :3]
        rec = 1.0 / (ds + 1e-8)
        w = rec / rec.sum(dim=2, keepdim=True)
        D2 = 16
        p2 = frs.normal(size=(B, x2.shape[1], D2)).astype(np.float32)
        interp = torch.sum(U.index_points(t(p2), di) * w.view(B, -1, 3, 1), dim=2)
        # mark rows whose rank-3/rank-4 distances tie (reference sort is unstable there)
        d4 = d.sort(dim=-1)[0][:, :, :4].numpy()
        tie = (d4[:, :, 0] == d4[:, :, 1]) | (d4[:, :, 1] == d4[:, :, 2]) | (d4[:, :, 2] == d4[:, :, 3])
        out["nn%d_idx" % lv] = di.numpy().astype(np.int16)
        out["nn%d_dist" % lv] = ds.numpy()
        out["nn%d_weight" % lv] = w.numpy()
        out["nn%d_tie" % lv] = tie
        out["nn%d_interp" % lv] = interp.numpy()
        oi, od, ow = orc.three_nn(x1.numpy(), x2.numpy())
        ok = ~tie
        assert np.array_equal(oi[ok], di.numpy()[ok]), (kind, lv, "nn idx")
        assert np.array_equal(od[ok], ds.numpy()[ok]), (kind, lv, "nn dist")
    out["nn_feat_seed"] = np.int64(seed + 1)
    # a small raw square_distance tile, bit-for-bit
    sd = U.square_distance(t(xyz_lv[1][:, :64]), t(xyz[:, :512])).numpy()
    out["sqdist_1024x4096_tile"] = sd
    assert np.array_equal(orc.square_distance(xyz_lv[1][:, :64], xyz[:, :512]).view(np.uint32), sd.view(np.uint32))
    save("geometry_" + kind, **out)


def fp_tie_masks(xyz0, starts):
    """Rows of each FP level (3,2,1,0) whose 4 smallest reference distances contain a tie: the
    reference's sort (pointnet2_utils.py:297) is unstable there, so its 3-NN pick is unpinned."""
    xyzs = [np.ascontiguousarray(xyz0)]
    for (npoint, _), st in zip(LEVELS, starts):
        f = orc.farthest_point_sample(xyzs[-1], npoint, st)       # == reference (asserted in gen_geometry)
        xyzs.append(orc.index_points(xyzs[-1], f))
    masks = {}
    for lv in (3, 2, 1, 0):
        d4 = U.square_distance(t(xyzs[lv]), t(xyzs[lv + 1])).sort(dim=-1)[0][:, :, :4].numpy()
        masks[lv] = (d4[:, :, 0] == d4[:, :, 1]) | (d4[:, :, 1] == d4[:, :, 2]) | (d4[:, :, 2] == d4[:, :, 3])
    return masks


def load_filled(model, num_classes, extra):
    shapes = orc.state_shapes(num_classes, extra)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict key order differs from the reference"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    filled = synth.fill_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    return filled


def draw_inputs(seed, B, N, C, kind, num_classes, upper_only):
    """Inputs for `seed`, moving on by +1000 while the reference's 3-NN pick is unpinned (a tie)
    at the FP levels that feed other points (upper_only) or at any level."""
    while True:
        blocks, labels, starts, cw = synth.draw_case(seed, B, N, C, kind, num_classes)
        ties = fp_tie_masks(blocks[:, :, :3], starts)
        levels = (3, 2, 1) if upper_only else (3, 2, 1, 0)
        if not any(ties[lv].any() for lv in levels):
            return seed, blocks, labels, starts, cw, ties
        print("   seed %d has a 3-NN tie, trying %d" % (seed, seed + 1000))
        seed += 1000


def gen_model_eval(kind, C, num_classes, seed, taps):
    B, N = 1, 4096
    seed, blocks, _, starts, _, ties = draw_inputs(seed, B, N, C, kind, num_classes, upper_only=True)
    model = M.get_model(num_classes, C - 6).eval()
    filled = load_filled(model, num_classes, C - 6)
    captured = {}
    hooks = []
    if taps:
        for name in ("sa1", "sa2", "sa3", "sa4"):
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, name=name: captured.__setitem__(name, o[1].detach().numpy().copy())))
        for name in ("fp4", "fp3", "fp2", "fp1"):
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, name=name: captured.__setitem__(name, o.detach().numpy().copy())))
    with torch.no_grad(), injected_fps_starts(starts):
        logp, l4 = model(t(blocks).permute(0, 2, 1))
    for h in hooks:
        h.remove()
    out = {"seed": np.int64(seed), "C": np.int64(C), "num_classes": np.int64(num_classes),
           "logp": logp.numpy(), "l4_points": l4.numpy()}
    for i, s in enumerate(starts, start=1):
        out["start%d" % i] = s
    for k, v in captured.items():
        out["tap_" + k] = v[:, :, ::8].copy() if k == "fp1" else v       # fp1 [1,128,4096] subsampled
    out["tie_points"] = ties[0]                                         # [B,N] output points with an unpinned 3-NN pick
    # oracle cross-check (not stored)
    net = orc.OracleNet(filled)
    with torch.no_grad():
        ologp, _ = net.forward(blocks.transpose(0, 2, 1), starts)
    ok = torch.from_numpy(~ties[0])
    err = float((ologp - logp).abs()[ok].max())
    print("   oracle vs reference max|dlogp| = %.3e (%d tie points excluded)" % (err, int(ties[0].sum())))
    assert err < 1e-5
    save("model_eval_%s_c%d_k%d" % (kind, C, num_classes), **out)


def gen_model_train(kind, C, num_classes, seed):
    B, N = 2, 4096
    seed, blocks, labels, starts, cw, _ = draw_inputs(seed, B, N, C, kind, num_classes, upper_only=False)
    model = M.get_model(num_classes, C - 6).train()
    filled = load_filled(model, num_classes, C - 6)
    model.drop1.p = 0.0                    # dropout off: its mask is torch-RNG dependent
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    crit = M.get_loss()
    opt.zero_grad()
    with injected_fps_starts(starts):
        pred, tf = model(t(blocks).permute(0, 2, 1))
    loss = crit(pred.contiguous().view(-1, num_classes), t(labels).view(-1, 1)[:, 0], tf, t(cw))
    loss.backward()
    names = ("sa1.mlp_convs.0.weight", "sa1.mlp_bns.0.weight", "sa2.mlp_convs.0.weight", "sa4.mlp_convs.2.bias",
             "fp4.mlp_convs.0.weight", "fp1.mlp_convs.0.weight", "fp1.mlp_bns.2.bias", "conv1.weight", "conv2.weight",
             "conv2.bias")
    params = dict(model.named_parameters())
    out = {"seed": np.int64(seed), "C": np.int64(C), "num_classes": np.int64(num_classes),
           "loss": np.float32(loss.item()), "class_weight": cw}
    for i, s in enumerate(starts, start=1):
        out["start%d" % i] = s
    for n in names:
        out["grad:" + n] = params[n].grad.numpy().copy()
    gn = {k: float(p.grad.norm()) for k, p in params.items()}
    out["grad_norm_keys"] = np.array(list(gn.keys()))
    out["grad_norm_vals"] = np.array(list(gn.values()), dtype=np.float32)
    opt.step()
    for n in names:
        out["adam:" + n] = params[n].detach().numpy().copy()
    sd = model.state_dict()
    for n in ("sa1.mlp_bns.0.running_mean", "sa1.mlp_bns.0.running_var", "fp1.mlp_bns.2.running_var", "bn1.running_mean"):
        out["buf:" + n] = sd[n].numpy().copy()
    # oracle cross-check
    net = orc.OracleNet(filled, dropout_p=0.0)
    oopt = orc.make_adam(net.parameters())
    oloss = net.train_step(blocks.transpose(0, 2, 1), labels, starts, oopt, cw)
    print("   oracle loss %.6f reference loss %.6f" % (oloss, loss.item()))
    assert abs(oloss - loss.item()) < 1e-4
    save("model_train_%s_c%d_k%d" % (kind, C, num_classes), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    gen_geometry("cube", 101)
    gen_geometry("facade", 102)
    gen_model_eval("cube", 9, 18, 201, taps=True)
    gen_model_eval("facade", 9, 18, 202, taps=False)
    gen_model_eval("cube", 6, 18, 203, taps=False)
    gen_model_eval("cube", 9, 8, 204, taps=False)
    gen_model_eval("facade", 6, 8, 205, taps=False)
    gen_model_train("cube", 9, 18, 301)
    gen_model_train("facade", 6, 8, 302)


if __name__ == "__main__":
    main()
