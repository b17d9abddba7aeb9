"""GPU: device vote aggregation (pn2_add_vote) against the oracle restatement of add_vote
(localfunctions.py:339-346), and a whole-scene inference round trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vote_pool_matches_add_vote():
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import ops, scene
    from oracle import scene_oracle as so
    rs = np.random.RandomState(9)
    B, N, P, C = 5, 1000, 3000, 18
    logp = rs.normal(size=(B, N, C)).astype(np.float32)
    logp[0, :50, 3] = logp[0, :50, 7] = 9.0                       # ties: the first maximum wins (torch.max)
    point_idx = rs.randint(0, P, size=(B, N))
    weight = rs.uniform(0.5, 2.0, size=(B, N)).astype(np.float32)
    weight[rs.rand(B, N) < 0.2] = 0.0
    weight[rs.rand(B, N) < 0.1] = np.inf
    pred = torch.from_numpy(logp).max(2)[1].numpy()               # the reference's host arg-max (:399)
    want = so.add_vote(np.zeros((P, C)), point_idx.astype(np.float64), pred, weight.astype(np.float64))
    dev = torch.device("cuda:0")
    pool = scene.VotePool(P, C, dev)
    pool.add(logp=torch.from_numpy(logp).to(dev), point_idx=torch.from_numpy(point_idx).to(dev),
             weight=torch.from_numpy(weight).to(dev))
    assert np.array_equal(pool.pool.cpu().numpy(), want.astype(np.int32))
    pool2 = scene.VotePool(P, C, dev)                             # explicit labels instead of logits
    pool2.add(pred_label=torch.from_numpy(pred).to(dev), point_idx=torch.from_numpy(point_idx).to(dev),
              weight=torch.from_numpy(weight).to(dev))
    assert np.array_equal(pool2.pool.cpu().numpy(), want.astype(np.int32))
    assert np.array_equal(pool.labels().cpu().numpy(), np.argmax(want, 1))
    ops.check_errors()
    pool.add(pred_label=torch.full((4,), C, device=dev), point_idx=torch.zeros(4, dtype=torch.long, device=dev))
    with pytest.raises(IndexError):
        ops.check_errors()


def test_infer_scene_round_trip(orc, synth):
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import scene
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from oracle import scene_oracle as so
    rs = np.random.RandomState(21)
    P, K = 9000, 8
    xyz = rs.uniform(0, 1, size=(P, 3)) * np.array([1.6, 1.2, 2.5]) + np.array([5.0, 7.0, 0.5])
    labels = rs.randint(0, K, size=(P,))
    rgb = [rs.randint(0, 256, size=(P,)).astype(np.float64) for _ in range(3)]
    np.random.seed(5)
    tiler = scene.SceneTiler(xyz, labels, rgb, ["red", "blue", "green"], block_points=2048)
    data, lab, wt, idx = tiler.tile()
    dev = torch.device("cuda:0")
    model = M.get_model(K, 3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(orc.state_shapes(K, 3)).items()})
    model = model.to(dev)
    pred = scene.infer_scene(model, data, idx, wt, P, K, batch_size=4).cpu().numpy()
    assert pred.shape == (P,) and pred.min() >= 0 and pred.max() < K
    # same votes through the reference's route: host arg-max + add_vote loop
    model.eval()
    pool = np.zeros((P, K))
    with torch.no_grad():
        for s in range(0, data.shape[0], 4):
            logp, _ = model(torch.as_tensor(data[s:s + 4], dtype=torch.float32, device=dev).transpose(2, 1))
            pool = so.add_vote(pool, idx[s:s + 4], logp.cpu().max(2)[1].numpy(), wt[s:s + 4])
    assert np.array_equal(pred, np.argmax(pool, 1))


def test_block_inferencer_matches_eager_forward(orc, synth, monkeypatch):
    """The replayed inference graph (next sub-batch's pyramid on a parallel branch, the first level's rows from the
    fused query launch, cached eval coefficients, no z of the pooled layers) gives the eager eval forward's bits, also
    for a short last sub-batch; infer_scene(graphs=True) votes like infer_scene()."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import scene
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda low, high, size, **kw: real_randint(0, 1, size, **kw))   # FPS starts
    K, B, N = 8, 4, 2048
    dev = torch.device("cuda:0")
    model = M.get_model(K, 3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(orc.state_shapes(K, 3)).items()})
    model = model.to(dev).eval()
    blocks = [synth.draw_case(300 + i, b, N, 9, kind, K)[0] for i, (b, kind) in enumerate(((4, "cube"), (4, "facade"), (3, "cube")))]
    xs = [torch.from_numpy(np.ascontiguousarray(b.transpose(0, 2, 1))).to(dev) for b in blocks]
    with torch.no_grad():
        want = [model(x)[0].clone() for x in xs[:2]]
        pad = torch.cat([xs[2], xs[2][-1:]])
        want.append(model(pad)[0][:3].clone())
    got = []
    engine = scene.BlockInferencer(model, B, 9, N)
    engine.run(xs, lambda i, logp: got.append(logp.clone()))
    engine.run(xs[:1], lambda i, logp: got.append(logp.clone()))          # a second scene through the same graph
    assert len(got) == 4
    for a, b in zip(got, want + want[:1]):
        assert a.shape == b.shape and torch.equal(a, b)
    # whole scene: same votes
    rs = np.random.RandomState(3)
    P = 7000
    xyz = rs.uniform(0, 1, size=(P, 3)) * np.array([1.6, 1.2, 2.5]) + np.array([5.0, 7.0, 0.5])
    labels = rs.randint(0, K, size=(P,))
    rgb = [rs.randint(0, 256, size=(P,)).astype(np.float64) for _ in range(3)]
    np.random.seed(5)
    data, lab, wt, idx = scene.SceneTiler(xyz, labels, rgb, ["red", "blue", "green"], block_points=N).tile()
    a = scene.infer_scene(model, data, idx, wt, P, K, batch_size=4)
    b = scene.infer_scene(model, data, idx, wt, P, K, batch_size=4, graphs=True)
    assert torch.equal(a, b)


def test_block_inferencer_kept_across_training_sees_the_trained_model(orc, synth, monkeypatch):
    """An engine captured before further training (the docstring recommends keeping one per model across scenes) replays the
    TRAINED model: run() refreshes the cached eval-mode BatchNorm coefficients in place.  Training moves weights (torch
    Adam: version counters bump) and running statistics (raw kernels: no version bump) -- both must arrive."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import scene
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda low, high, size, **kw: real_randint(0, 1, size, **kw))   # FPS starts
    K, B, N = 8, 2, 2048
    dev = torch.device("cuda:0")
    model = M.get_model(K, 3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(orc.state_shapes(K, 3)).items()})
    model = model.to(dev).eval()
    blocks, labels, _, _ = synth.draw_case(411, B, N, 9, "cube", K)
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(labels).to(dev).view(-1)
    engine = scene.BlockInferencer(model, B, 9, N)
    first = []
    engine.run([x], lambda i, logp: first.append(logp.clone()))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        logp, tf = model(x)
        M.get_loss()(logp.reshape(-1, K), y, tf, None).backward()
        opt.step()
    model.eval()
    got = []
    engine.run([x], lambda i, logp: got.append(logp.clone()))     # BEFORE any eager eval forward could refresh the cache
    with torch.no_grad():
        want = model(x)[0].clone()
    assert not torch.equal(want, first[0])                 # the model did move
    assert torch.equal(got[0], want)
