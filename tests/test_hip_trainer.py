"""GPU: the captured-graph / geometry-prefetch training step must train like the plain eager
step, and a precomputed geometry pyramid must reproduce the in-forward one exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B, N, C, K = 4, 2048, 9, 18


def _setup(monkeypatch):
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from oracle import pn2_oracle as orc
    # FPS starts are drawn with torch.randint (reference pointnet2_utils.py:75); pin them to 0 so
    # that every mode samples the same pyramid regardless of the order RNG calls are issued in
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda low, high, size, **kw: real_randint(0, 1, size, **kw))
    dev = torch.device("cuda:0")
    data = [synth.draw_case(77, B, N, C, "cube", K), synth.draw_case(78, B, N, C, "facade", K)]
    xs = [torch.from_numpy(np.ascontiguousarray(d[0].transpose(0, 2, 1))).to(dev) for d in data]
    ys = [torch.from_numpy(d[1]).to(dev) for d in data]
    cw = torch.from_numpy(data[0][3]).to(dev)

    def fresh_model():
        model = M.get_model(K, C - 6)
        filled = synth.fill_state_dict(orc.state_shapes(K, C - 6))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        model = model.to(dev)
        model.drop1.p = 0.0
        return model
    return torch, xs, ys, cw, fresh_model


def _train(torch, xs, ys, cw, model, steps, **mode):
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    tr = SemSegTrainer(model, class_weight=cw, graph_warmup=0, **mode)
    losses = [float(tr.step(xs[i % 2], ys[i % 2], xs[(i + 1) % 2])) for i in range(steps)]
    torch.cuda.synchronize()
    return np.array(losses)


def test_graph_and_prefetch_steps_match_eager(monkeypatch):
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    _train(torch, xs, ys, cw, fresh_model(), 2, graphs=False, prefetch_geometry=False)   # lazy inits, MIOpen find
    steps = 5
    ref = _train(torch, xs, ys, cw, fresh_model(), steps, graphs=False, prefetch_geometry=False)
    assert np.isfinite(ref).all() and ref[-1] < ref[0]
    for mode in (dict(graphs=False, prefetch_geometry=True), dict(graphs=True, prefetch_geometry=False),
                 dict(graphs=True, prefetch_geometry=True)):
        got = _train(torch, xs, ys, cw, fresh_model(), steps, **mode)
        # Identical kernels and inputs.  The first step must agree to rounding; after that the
        # trajectories drift apart chaotically (float-atomic scatter-adds reorder sums, and Adam
        # turns the sign of a noise-level gradient -- conv biases under BatchNorm -- into a full
        # +-lr step), in either mode and between two runs of the same mode.
        assert abs(got[0] - ref[0]) <= 1e-4, mode
        np.testing.assert_allclose(got, ref, rtol=2e-2, err_msg=str(mode))
        assert got[-1] < got[0], mode


# the captured step's forms: module switches of train.py (A/B evidence in DESIGN.md 6); every one that ships is run here
STEP_FORMS = {"two graphs": {"_ALTERNATE_STEP_GRAPHS": False},
              "forked graph": {"_SEPARATE_GEOMETRY_GRAPH": False},
              "forked graph, hand-over on the side branch": {"_SEPARATE_GEOMETRY_GRAPH": False, "_HANDOVER_ON_MAIN": False},
              "alternating step graphs": {"_ALTERNATE_STEP_GRAPHS": True},
              "geometry graph enqueued ahead, behind a cross-stream wait": {"_LATE_SIDE_ENQUEUE": False},
              "small tensors packed through a temporary": {"_PACK_IN_PLACE": False}}


@pytest.mark.parametrize("form", sorted(STEP_FORMS))
def test_every_replay_reads_the_pyramid_of_the_batch_it_trains_on(monkeypatch, form):
    """Graph + prefetch on alternating cube / facade batches: the pyramid tensors a replay is about to read (cloned by a
    hook right in front of it) equal compute_geometry() of THAT batch bit for bit -- the prefetched one when the batch was
    announced, a freshly computed one when it was not (a step on another batch than the one announced; the same batch
    twice in a row).  A stale or foreign pyramid cannot hide behind a loss tolerance here.  Every form of the step that ships behind a switch."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd import train as T
    for k, v in STEP_FORMS[form].items():
        monkeypatch.setattr(T, k, v)
    model = fresh_model()
    want = []
    with torch.no_grad():
        for x in xs:                                   # the pyramid is a function of the coordinates only (FPS starts pinned)
            prepared = model.prepare_input(x, None)
            want.append(model.compute_geometry(prepared=prepared, group_first=True) + [prepared[0], prepared[1]])
    tr = T.SemSegTrainer(model, class_weight=cw, graphs=True, prefetch_geometry=True, graph_warmup=0)
    taps = []
    tr._tap = taps.append
    # (batch trained on, batch announced as the next one)
    plan = [(0, 1), (1, 0), (0, 1), (1, 1), (1, 0), (1, 0), (0, None), (0, 1), (1, 0)]    # 5: announced 0, trains 1; 6: no announcement
    losses = []
    for b, nxt in plan:
        losses.append(tr.step(xs[b], ys[b], None if nxt is None else xs[nxt]))
    torch.cuda.synchronize()
    forked = form.startswith("forked")
    assert (tr._alt is not None) == (not forked and STEP_FORMS[form].get("_ALTERNATE_STEP_GRAPHS", True))
    assert (tr._g_geo is not None) == (not forked)
    assert len(taps) == len(plan)
    for i, ((b, _), read) in enumerate(zip(plan, taps)):
        assert len(read) == len(want[b])
        for j, (a, e) in enumerate(zip(read, want[b])):
            assert (a is None) == (e is None), (form, i, j)
            if a is not None:
                assert a.shape == e.shape and torch.equal(a, e), "%s: step %d read a pyramid that is not batch %d's (tensor %d)" % (form, i, b, j)
    assert all(torch.isfinite(l) for l in losses)


def test_precomputed_geometry_is_identical(monkeypatch):
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    model = fresh_model().eval()
    with torch.no_grad():
        a, a4 = model(xs[1])
        geo = model.compute_geometry(xs[1])
        b, b4 = model(xs[1], geometry=geo)
        # ... and with the first level's grouped rows built by the launch that finds its indices (the trainer's form)
        prepared = model.prepare_input(xs[1])
        geo2 = model.compute_geometry(prepared=prepared, group_first=True)
        assert len(geo2) == 33 and geo2[32] is not None and geo2[32].shape == (B, 1024, 32, 12)
        c, c4 = model(xs[1], geometry=geo2, prepared=prepared)
        from khairil_tum_facade_semantic_segmentation_amd import ops
        assert torch.equal(geo2[32], ops.group_points(prepared[1], geo2[0], prepared[0], geo2[1], pad_to=4))
    assert torch.equal(a, b) and torch.equal(a4, b4)
    assert torch.equal(a, c) and torch.equal(a4, c4)


def test_skip_gradient_summed_inside_the_grouping_backward(monkeypatch):
    """With a precomputed pyramid the skip connections are routed through the grouping op's second output (their
    gradient meets the grouping gradient inside the scatter): same parameter gradients as the plain wiring."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    grads = []
    for flag in (False, True):
        monkeypatch.setattr(M, "_SKIP_IN_SCATTER", flag)
        model = fresh_model().train()
        geo = model.compute_geometry(xs[0])
        logp, _ = model(xs[0], geometry=geo)
        loss = M.get_loss()(logp.reshape(-1, K), ys[0].reshape(-1), None, cw)
        loss.backward()
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        s = float(a.abs().max()) + 1e-8
        if k.endswith("convs.0.bias") or ".bias" in k and "mlp_convs" in k:
            continue                                  # conv biases under train-mode BatchNorm: pure rounding noise
        assert float((a - b).abs().max()) <= 2e-4 * s + 1e-7, k


def test_prepare_captures_without_touching_the_model(monkeypatch):
    """SemSegTrainer.prepare() (graph capture before the first collective) must leave parameters, BatchNorm
    buffers and the Adam state as if nothing had run, and the steps after it must train like eager steps."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    steps = 4
    ref = _train(torch, xs, ys, cw, fresh_model(), steps, graphs=False, prefetch_geometry=False)
    model = fresh_model()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = SemSegTrainer(model, class_weight=cw, graphs=True, prefetch_geometry=True, graph_warmup=2)
    tr.prepare(xs[0], ys[0])
    torch.cuda.synchronize()
    assert tr._g_fwd_bwd is not None
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    for v in (tr.flat_adam.exp_avg, tr.flat_adam.exp_avg_sq, tr.flat_adam.state):
        assert float(v.abs().max()) == 0.0
    got = np.array([float(tr.step(xs[i % 2], ys[i % 2], xs[(i + 1) % 2])) for i in range(steps)])
    assert abs(got[0] - ref[0]) <= 1e-4
    np.testing.assert_allclose(got, ref, rtol=2e-2)
    assert int(model.sa1.mlp_bns[0].num_batches_tracked) == steps


def test_flat_adam_matches_torch_adam():
    """pn2_adam_step (one pass over the flat parameter buffer) against torch.optim.Adam with the reference's
    settings (sem_seg_training.py:576-582): three steps, L2 weight decay, bias correction, a learning-rate change
    and a gradient scale."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd.train import FlatAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 67, 1), (64,), (13, 128, 1), (5,), (128, 128, 1)]
    ref_p = [torch.nn.Parameter(torch.randn(sh, generator=g).double()) for sh in shapes]
    our_p = [torch.nn.Parameter(p.detach().float().cuda()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    ours = FlatAdam(our_p, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    for step in range(3):
        grads = [torch.randn(sh, generator=g) * (10.0 ** (step - 1)) for sh in shapes]
        if step == 2:
            for grp in ref.param_groups:
                grp["lr"] = 7e-4
            ours.set_lr(7e-4)
        for p, gr in zip(ref_p, grads):
            p.grad = gr.double() * 0.5
        ref.step()
        flat = torch.cat([gr.reshape(-1) for gr in grads]).cuda()
        ours.step(flat, grad_scale=0.5)
        for a, b in zip(our_p, ref_p):
            torch.testing.assert_close(a.detach().cpu().double(), b.detach(), rtol=2e-6, atol=2e-7)
    assert float(ours.state[0]) == 3.0


def test_adam_reads_gradients_where_backward_left_them():
    """pn2_adam_step_scattered (one gradient pointer per parameter tensor, a missing gradient = zeros; tensor sizes
    that are no multiples of 4, so float4 groups straddle tensors) gives the bits of the packed update."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd.train import FlatAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 67, 1), (63,), (13, 128, 1), (5,), (1,), (128, 128, 1), (7, 3)]
    init = [torch.randn(sh, generator=g) for sh in shapes]
    a_p = [torch.nn.Parameter(p.clone().cuda()) for p in init]
    b_p = [torch.nn.Parameter(p.clone().cuda()) for p in init]
    a = FlatAdam(a_p, lr=1e-3, weight_decay=1e-4)
    b = FlatAdam(b_p, lr=1e-3, weight_decay=1e-4)
    for step in range(3):
        grads = [torch.randn(sh, generator=g).cuda() for sh in shapes]
        if step == 1:
            grads[3] = None                       # a parameter the loss did not reach
        flat = torch.cat([(gr if gr is not None else torch.zeros(sh, device="cuda")).reshape(-1)
                          for gr, sh in zip(grads, shapes)])
        a.step(flat, grad_scale=0.25)
        assert b.step_scattered(grads, grad_scale=0.25)
        assert torch.equal(a.flat, b.flat) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert float(b.state[0]) == 3.0


def _dp_worker(rank, world, port, out):
    """One data-parallel rank on the (shared) GPU: gloo process group, captured graphs, own batch."""
    import os
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from khairil_tum_facade_semantic_segmentation_amd import synth
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    from oracle import pn2_oracle as orc
    real_randint = torch.randint
    torch.randint = lambda low, high, size, **kw: real_randint(0, 1, size, **kw)      # same FPS starts everywhere
    dev = torch.device("cuda:0")
    d = synth.draw_case(100 + rank, 2, 1024, 9, "cube", 18)                          # every rank its own blocks
    x = torch.from_numpy(np.ascontiguousarray(d[0].transpose(0, 2, 1))).to(dev)
    y = torch.from_numpy(d[1]).to(dev)
    model = M.get_model(18, 3)
    filled = synth.fill_state_dict(orc.state_shapes(18, 3))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.to(dev)
    model.drop1.p = 0.0
    tr = SemSegTrainer(model, class_weight=torch.ones(18, device=dev), graphs=True, prefetch_geometry=True, graph_warmup=1)
    tr.prepare(x, y)                      # capture before the first collective
    tr.broadcast_parameters()
    losses = [float(tr.step(x, y)) for _ in range(3)]
    torch.cuda.synchronize()
    assert tr._g_opt is not None          # the two-graph exchange path was taken
    flat = tr.flat_adam.flat.detach().cpu().numpy()
    np.save(os.path.join(out, "params_rank%d.npy" % rank), flat)
    np.save(os.path.join(out, "loss_rank%d.npy" % rank), np.array(losses))
    dist.destroy_process_group()


def test_two_rank_exchange_on_one_gpu(tmp_path):
    """world_size 2 (gloo, both ranks on this GPU): graph-captured forward/backward, all-reduce of the packed
    gradients between the graphs, flat Adam with the 1/world folded in.  Replicas must stay bit-identical."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_rank0.npy"), np.load(tmp_path / "params_rank1.npy")
    assert np.array_equal(p0, p1)
    l0, l1 = np.load(tmp_path / "loss_rank0.npy"), np.load(tmp_path / "loss_rank1.npy")
    assert np.isfinite(l0).all() and np.isfinite(l1).all() and not np.array_equal(l0, l1)


def _dp_epoch_worker(rank, world, port, out):
    """One rank of a two-rank train_epoch on the shared GPU (gloo): device sampler, augmentation, captured step."""
    import os
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    from make_golden_scene import make_scene
    from khairil_tum_facade_semantic_segmentation_amd import scene
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_sem_seg as M
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer, draw_batch, train_epoch
    K = 8
    rooms = [make_scene(31, 60000), make_scene(32, 40000, extent=(1.6, 1.4, 2.5))]
    samplers = [scene.DeviceBlockSampler(r[0], r[1], r[2], ["red", "blue", "green"]) for r in rooms]
    torch.manual_seed(0)
    model = M.get_model(K, 3).cuda()
    tr = SemSegTrainer(model, class_weight=torch.ones(K, device="cuda"), graphs=True, prefetch_geometry=True, graph_warmup=1,
                       augment=False, metrics=True)
    first = draw_batch(samplers, 4, 3, 0, 0, rank)              # what train_epoch draws for step 0 on this rank
    tr.prepare(first[0].contiguous(), first[1])
    tr.broadcast_parameters()
    e = train_epoch(tr, samplers, 0, steps=3, batch_size=4, seed=3)    # rank taken from the process group
    torch.cuda.synchronize()
    np.save(os.path.join(out, "params_rank%d.npy" % rank), tr.flat_adam.flat.detach().cpu().numpy())
    np.save(os.path.join(out, "first_rank%d.npy" % rank), first[0].cpu().numpy())
    np.save(os.path.join(out, "loss_rank%d.npy" % rank), np.array([e["loss"], e["seen"]]))
    dist.destroy_process_group()


def test_two_rank_epoch_draws_different_blocks(tmp_path):
    """train_epoch under a process group: the rank enters the sampler seeds, so the replicas train on
    DIFFERENT blocks (global batch = world x per-rank batch), and after the gradient all-reduce they hold identical
    parameters."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_epoch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_rank0.npy"), np.load(tmp_path / "params_rank1.npy")
    assert np.array_equal(p0, p1)
    f0, f1 = np.load(tmp_path / "first_rank0.npy"), np.load(tmp_path / "first_rank1.npy")
    assert f0.shape == f1.shape and not np.array_equal(f0, f1)
    l0, l1 = np.load(tmp_path / "loss_rank0.npy"), np.load(tmp_path / "loss_rank1.npy")
    assert np.isfinite(l0).all() and np.isfinite(l1).all() and l0[0] != l1[0] and l0[1] == l1[1] == 3 * 4 * 4096


def test_schedules_reach_a_captured_step(monkeypatch):
    """The reference loop resets the learning rate and every BatchNorm momentum each epoch
    (localfunctions.py:187-195).  After the step has been captured into hipGraphs both must still take effect:
    the replayed launches read them from device words.  One step in each mode from the same state, after the
    schedule moved: identical running statistics (they depend on the momentum only) and parameters."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    results = {}
    for graphs in (False, True):
        model = fresh_model()
        tr = SemSegTrainer(model, class_weight=cw, graphs=graphs, prefetch_geometry=False, graph_warmup=0, lr=1e-3)
        tr.step(xs[0], ys[0])                              # graph mode: captures here with momentum 0.1, lr 1e-3
        tr.step(xs[0], ys[0])
        tr.set_lr(2.5e-4)
        tr.set_bn_momentum(0.01)
        assert model.sa2.mlp_bns[1].momentum == 0.01
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        tr.step(xs[1], ys[1])
        torch.cuda.synchronize()
        results[graphs] = (before, {k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in results[False][1]:
        a0, a1 = results[False][0][k].double(), results[False][1][k].double()
        b0, b1 = results[True][0][k].double(), results[True][1][k].double()
        if "running_" in k:
            # new = (1 - m) * old + m * batch  ->  the step's relative move identifies m (0.01, not the captured 0.1)
            move_e, move_g = (a1 - a0), (b1 - b0)
            scale = float(move_e.abs().max()) + 1e-12
            assert float((move_e - move_g).abs().max()) <= 0.05 * scale + 1e-7, k
        elif "num_batches" in k:
            assert torch.equal(a1, b1)
    # the parameter update of that step scales with the learning rate: compare update norms
    def upd(res):
        return sum(float((res[1][k].double() - res[0][k].double()).pow(2).sum()) for k in res[1]
                   if "running_" not in k and "num_batches" not in k) ** 0.5
    ue, ug = upd(results[False]), upd(results[True])
    assert abs(ue - ug) <= 0.05 * ue, (ue, ug)
    # and momentum really moved: with the captured 0.1 the running mean would have moved ~10x as far
    k = "sa1.mlp_bns.0.running_mean"
    tr_model = fresh_model()
    tr2 = SemSegTrainer(tr_model, class_weight=cw, graphs=True, prefetch_geometry=False, graph_warmup=0, lr=1e-3)
    tr2.step(xs[0], ys[0]); tr2.step(xs[0], ys[0])
    b0 = tr_model.state_dict()[k].detach().clone()
    tr2.step(xs[1], ys[1])
    torch.cuda.synchronize()
    big = float((tr_model.state_dict()[k] - b0).abs().max())
    small = float((results[True][1][k] - results[True][0][k]).abs().max())
    assert small < 0.3 * big, (small, big)


def test_unannounced_batch_gets_its_own_pyramid(monkeypatch):
    """With geometry prefetch a step groups with the pyramid computed one call earlier.  A caller that does not
    announce the next batch (next_blocks_cf) and then trains on a different one must not get the previous batch's
    indices: first-step losses on alternating batches equal the eager / no-prefetch ones."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer

    def run(**mode):
        tr = SemSegTrainer(fresh_model(), class_weight=cw, graph_warmup=0, **mode)
        out = [float(tr.step(xs[i % 2], ys[i % 2])) for i in range(4)]      # no next_blocks_cf
        torch.cuda.synchronize()
        return np.array(out)
    ref = run(graphs=False, prefetch_geometry=False)
    for mode in (dict(graphs=False, prefetch_geometry=True), dict(graphs=True, prefetch_geometry=True)):
        got = run(**mode)
        assert abs(got[0] - ref[0]) <= 1e-4 and abs(got[1] - ref[1]) <= 5e-3, (mode, got, ref)
        np.testing.assert_allclose(got, ref, rtol=2e-2, err_msg=str(mode))


def test_bn_momentum_none_is_the_cumulative_average():
    """nn.BatchNorm(momentum=None): running = cumulative average over the batches seen."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import mlp
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv1d(8, 16, 1).to(dev)
    bn = torch.nn.BatchNorm1d(16, momentum=None).to(dev)
    ref_bn = torch.nn.BatchNorm1d(16, momentum=None).to(dev)
    for step in range(3):
        x = torch.randn(4096, 8, device=dev) * (1.0 + step) + step
        y = mlp.mlp_stack(x, None, [conv], [bn], 0)
        z = torch.nn.functional.conv1d(x.t().unsqueeze(0), conv.weight, conv.bias)
        yr = torch.relu(ref_bn(z)).squeeze(0).t()
        assert float((y - yr).abs().max()) <= 1e-4
    assert int(bn.num_batches_tracked) == 3
    assert float((bn.running_mean - ref_bn.running_mean).abs().max()) <= 1e-5
    assert float((bn.running_var - ref_bn.running_var).abs().max()) <= 1e-4 * float(ref_bn.running_var.abs().max())


@pytest.mark.parametrize("graphs", (False, True))
def test_eval_after_training_uses_the_trained_statistics(monkeypatch, graphs):
    """eval -> train -> eval (the reference loop validates after every epoch, localfunctions.py:243-263): BatchNorm running
    statistics and affine parameters are rewritten through raw pointers / graph replays, which bump no version counter; the
    second eval must NOT reuse the first one's cached coefficients.  Reference: a fresh model loaded
    with the trained state."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    model = fresh_model()
    with torch.no_grad():
        before, _ = model.eval()(xs[0])
        before = before.clone()
    tr = SemSegTrainer(model, class_weight=cw, graphs=graphs, prefetch_geometry=graphs, graph_warmup=0)
    for i in range(3):
        tr.step(xs[i % 2], ys[i % 2], xs[(i + 1) % 2])
    with torch.no_grad():
        after, _ = model.eval()(xs[0])
        twin = fresh_model()
        twin.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        want, _ = twin.eval()(xs[0])
    assert torch.equal(after, want)
    assert (after - before).abs().max() > 1e-3              # training did move the output


def test_prefetch_identity_survives_address_reuse(monkeypatch):
    """The pyramid prefetched for an announced batch belongs to THAT tensor: a different batch that the allocator places at
    the same address (both written by raw kernels: version 0) must get its own pyramid."""
    torch, xs, ys, cw, fresh_model = _setup(monkeypatch)
    from khairil_tum_facade_semantic_segmentation_amd.train import SemSegTrainer
    tr = SemSegTrainer(fresh_model(), class_weight=cw, graphs=False, prefetch_geometry=True, graph_warmup=0)
    a = xs[0].clone()
    tr.step(a, ys[0], a)                                    # announces `a` as its own successor
    ptr = a.data_ptr()
    held = tr._geo_next_src[0]
    assert held is a
    del a                                                   # the trainer still holds it: the address cannot be recycled
    b = xs[1].clone()
    assert b.data_ptr() != ptr
    ref = SemSegTrainer(fresh_model(), class_weight=cw, graphs=False, prefetch_geometry=False, graph_warmup=0)
    ref.step(xs[0], ys[0])
    want = float(ref.step(xs[1], ys[1]))
    got = float(tr.step(b, ys[1]))
    assert abs(got - want) <= 5e-3
    tr.drop_prefetched()
    assert tr._geo_next is None and tr._geo_next_src is None


def test_ball_plan_of_another_cloud_is_not_used():
    """ops.BallPlan remembers the tensors it was built from: passing it with another cloud of the same shape rebuilds the
    plan instead of returning the first cloud's neighbours."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    x1 = torch.rand(4, 2048, 3, generator=g).cuda()
    x2 = torch.rand(4, 2048, 3, generator=g).cuda()
    f1, f2 = torch.rand(4, 2048, 9, generator=g).cuda(), torch.rand(4, 2048, 9, generator=g).cuda()
    _, c1, plan = ops.farthest_point_sample_plan(x1, 1024, 0.1, 9, torch.zeros(4, dtype=torch.long, device="cuda"))
    _, c2 = ops.farthest_point_sample_with_xyz(x2, 1024, torch.zeros(4, dtype=torch.long, device="cuda"))
    want_idx, want_rows = ops.ball_query_group(0.1, 32, x2, c2, f2)
    got_idx, got_rows = ops.ball_query_group(0.1, 32, x2, c2, f2, plan=plan)        # foreign plan: ignored
    assert torch.equal(got_idx, want_idx) and torch.equal(got_rows, want_rows)
    i1, r1 = ops.ball_query_group(0.1, 32, x1, c1, f1, plan=plan)
    i1b, r1b = ops.ball_query_group(0.1, 32, x1, c1, f2, plan=plan)                 # same geometry, other features: rows re-packed
    assert torch.equal(i1, i1b) and not torch.equal(r1, r1b)
    assert torch.equal(r1b[..., 3:], ops.index_points(f2, i1))
