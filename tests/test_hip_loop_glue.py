"""GPU: the loop glue of SURVEY.md 8f row 4 -- the rotate-z augmentation inside the input-preparation kernel and the
accuracy / IoU counters kept on the device -- against the formulas of the reference loop (provider.py:66-84,
localfunctions.py:214-223, 271-289), and wired into SemSegTrainer.step in every mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rotate_ref(blocks, angles):
    """provider.rotate_point_cloud_z with given angles: float64 matrix product, stored as float32"""
    out = blocks.copy()
    for k, a in enumerate(angles):
        c, s = np.cos(a), np.sin(a)
        rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]])
        out[k, :, :3] = np.dot(blocks[k, :, :3].reshape(-1, 3), rot).astype(np.float32)
    return out


def test_input_blocks_rotates_and_lays_out():
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import ops
    rs = np.random.RandomState(0)
    B, N, C = 5, 777, 9
    blocks = rs.normal(size=(B, N, C)).astype(np.float32)
    angles = rs.uniform(0, 2 * np.pi, size=B).astype(np.float32)
    want = _rotate_ref(blocks, angles.astype(np.float64))
    for channel_first in (True, False):
        x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1)) if channel_first else blocks).cuda()
        pts, xyz = ops.input_blocks(x, channel_first, torch.from_numpy(angles).cuda())
        assert np.abs(pts.cpu().numpy() - want).max() <= 2e-6
        assert np.array_equal(xyz.cpu().numpy(), pts.cpu().numpy()[:, :, :3])
        pts0, xyz0 = ops.input_blocks(x, channel_first, None)          # no rotation: an exact re-layout
        assert np.array_equal(pts0.cpu().numpy(), blocks) and np.array_equal(xyz0.cpu().numpy(), blocks[:, :, :3])


def test_seg_metrics_match_the_loop_formulas():
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import ops
    rs = np.random.RandomState(1)
    C = 18
    m = ops.SegMetrics(C, torch.device("cuda:0"))
    tc = ts = 0
    seen = np.zeros(C); corr = np.zeros(C); union = np.zeros(C)
    for it in range(3):
        logp = rs.normal(size=(4, 1000, C)).astype(np.float32)
        logp[0, :50] = logp[0, :50].round()                              # ties: the first maximum wins
        lab = rs.randint(0, C, size=(4, 1000))
        m.add(torch.from_numpy(logp).cuda(), torch.from_numpy(lab).cuda())
        pred = logp.argmax(2)                                            # localfunctions.py:271
        tc += (pred == lab).sum(); ts += lab.size
        for l in range(C):                                               # :278-281
            seen[l] += (lab == l).sum()
            corr[l] += ((pred == l) & (lab == l)).sum()
            union[l] += ((pred == l) | (lab == l)).sum()
    r = m.read()
    assert (r["correct"], r["seen"]) == (tc, ts)
    assert np.array_equal(r["class_seen"], seen) and np.array_equal(r["class_correct"], corr)
    assert np.array_equal(r["class_union"], union)
    assert abs(r["mIoU"] - np.mean(corr / (union + 1e-6))) < 1e-12
    m.reset()
    assert m.read()["seen"] == 0


@pytest.mark.parametrize("mode", (dict(graphs=False, prefetch_geometry=False), dict(graphs=False, prefetch_geometry=True),
                                  dict(graphs=True, prefetch_geometry=True)))
def test_trainer_augments_and_counts_on_the_device(monkeypatch, mode):
    """A step with augment=True must train on exactly the rotated batch (same loss as a step on a batch rotated
    beforehand with the reference formula), and metrics=True must count every point of every step."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    from oracle import pn2_oracle as orc
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda low, high, size, **kw: real_randint(0, 1, size, **kw))   # FPS starts = 0
    B, N, C, K = 4, 2048, 9, 18
    dev = torch.device("cuda:0")
    blocks, labels, _, cw = synth.draw_case(91, B, N, C, "cube", K)
    angles = np.array([0.3, 1.7, 4.0, 5.9], np.float32)
    rotated = _rotate_ref(blocks, angles.astype(np.float64))
    y = torch.from_numpy(labels).to(dev)
    cwt = torch.from_numpy(cw).to(dev)

    def fresh():
        model = M.get_model(K, C - 6)
        filled = synth.fill_state_dict(orc.state_shapes(K, C - 6))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        model = model.to(dev)
        model.drop1.p = 0.0
        return model
    ref = SemSegTrainer(fresh(), class_weight=cwt, graph_warmup=0)
    xr = torch.from_numpy(np.ascontiguousarray(rotated.transpose(0, 2, 1))).to(dev)
    want = [float(ref.step(xr, y)) for _ in range(3)]
    tr = SemSegTrainer(fresh(), class_weight=cwt, graph_warmup=0, augment=True, metrics=True, **mode)
    ang_dev = torch.from_numpy(angles).to(dev)              # resident: the hook runs inside the graph capture too
    monkeypatch.setattr(tr, "_draw_angles", lambda b: ang_dev.clone())
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    got = [float(tr.step(x, y, x)) for _ in range(3)]
    torch.cuda.synchronize()
    assert abs(got[0] - want[0]) <= 1e-4, (got, want)
    np.testing.assert_allclose(got, want, rtol=2e-2)
    r = tr.metrics.read()
    assert r["seen"] == 3 * B * N and 0 <= r["correct"] <= r["seen"]
    assert r["class_seen"].sum() == r["seen"]


def test_epoch_runs_entirely_on_the_device():
    """scene.DeviceBlockSampler -> SemSegTrainer.step(augment, metrics, prefetch, graphs) under the reference's
    epoch schedule: two short epochs, losses finite and falling, every point counted, schedule applied."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from make_golden_scene import make_scene
    from khairil_tum_facade_semantic_segmentation_amd import scene, synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer, train_epoch, epoch_schedule
    assert epoch_schedule(0) == (1e-3, 0.1) and epoch_schedule(25) == (1e-3 * 0.7 ** 2, 0.025) and epoch_schedule(400)[1] == 0.01
    K = 8
    rooms = [make_scene(31, 60000), make_scene(32, 40000, extent=(1.6, 1.4, 2.5))]
    samplers = [scene.DeviceBlockSampler(r[0], r[1], r[2], ["red", "blue", "green"]) for r in rooms]
    model = M.get_model(K, 3).cuda()
    tr = SemSegTrainer(model, class_weight=torch.ones(K, device="cuda"), graphs=True, prefetch_geometry=True, graph_warmup=1,
                       augment=True, metrics=True)
    e0 = train_epoch(tr, samplers, 0, steps=4, batch_size=4, seed=3)
    e1 = train_epoch(tr, samplers, 10, steps=4, batch_size=4, seed=3)
    assert np.isfinite([e0["loss"], e1["loss"]]).all() and e1["loss"] < e0["loss"]
    assert e0["seen"] == 4 * 4 * 4096 and e1["seen"] == 4 * 4 * 4096
    assert (e1["lr"], e1["bn_momentum"]) == epoch_schedule(10)
    assert model.sa1.mlp_bns[0].momentum == e1["bn_momentum"]
