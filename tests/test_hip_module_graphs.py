"""GPU: the drop-in modules replaying captured graphs (khairil_tum-facade_semantic_segmentation_amd/graphed.py) against the
SAME modules launching eagerly -- the reference's wiring (tests/dropin_wiring.py), nothing changed in the caller.  The graphed
mode is on by default; `graphed.ENABLED = False` is the eager control.  Reference: models/pointnet2_sem_seg.py:22-40 (the
caller), localfunctions.py:203-218 (its loop: zero_grad, forward, nll_loss, backward, Adam)."""
import numpy as np
import pytest

from dropin_wiring import build, loss_fn

pytestmark = pytest.mark.gpu


def _setup(synth, orc, K=8, C=9, B=2, N=2048, kind="cube", seed=301):
    import torch
    from khairil_tum_facade_semantic_segmentation_amd import graphed
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    blocks, labels, starts, cw = synth.draw_case(seed, B, N, C, kind, K)
    model = build(U, K, C - 6)
    filled = synth.fill_state_dict(orc.state_shapes(K, C - 6))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    model = model.cuda()
    model.drop1.p = 0.0
    x = torch.from_numpy(np.ascontiguousarray(blocks.transpose(0, 2, 1))).cuda()
    y = torch.from_numpy(labels).cuda().view(-1)
    return torch, graphed, U, model, x, y, starts, K


def _close(a, b, rel=1e-5):
    return float((a - b).abs().max()) <= rel * float(b.abs().max()) + 2e-7      # 2e-7: gradients that are exactly 0 in
    #                                                                            exact arithmetic carry atomics' noise


def _train_pass(torch, U, model, x, y, starts, K, zero=True):
    if zero:
        model.zero_grad(set_to_none=True)
    with U.fps_starts(starts):
        pred, l4 = model(x)
    loss = loss_fn(pred.contiguous().view(-1, K), y, None)
    loss.backward()
    return (pred.detach().clone(), l4.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()},
            {k: b.clone() for k, b in model.named_buffers()})


def _same(torch, got, want, bitwise=True):
    if bitwise:
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        for k in want[3]:
            assert torch.equal(got[3][k], want[3][k]), k           # BatchNorm running statistics, batch counters
    else:
        assert _close(got[0], want[0], 1e-4) and _close(got[1], want[1], 1e-4)
    for k in want[2]:
        if k.endswith(".bias") and ("mlp_convs" in k or k == "conv1.bias"):
            continue                                               # exact gradient 0 under BatchNorm: atomics' rounding noise
        assert _close(got[2][k], want[2][k], 1e-5 if bitwise else 2e-3), k


def test_replayed_training_passes_equal_eager_ones(orc, synth, monkeypatch):
    """Six passes (no optimizer: BatchNorm statistics move, weights do not): eager, then the same six from the same state with
    the graphed mode on -- two warm-up passes, the capture, replays; outputs and running statistics bit for bit, gradients to
    the float atomics' order noise.  Every module replayed; the module whose output the caller keeps bound (`l4`, the
    reference loop's trans_feat) alternates between instances instead of overwriting it."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    monkeypatch.setattr(graphed, "ENABLED", False)
    eager = [_train_pass(torch, U, model, x, y, starts, K) for _ in range(6)]
    model.load_state_dict(state)
    monkeypatch.setattr(graphed, "ENABLED", True)
    before = dict(graphed.stats)
    held = []
    for i in range(6):
        got = _train_pass(torch, U, model, x, y, starts, K)
        _same(torch, got, eager[i])
        held.append(got)
    assert graphed.stats["captures"] - before["captures"] >= 8
    assert graphed.stats["replays"] - before["replays"] >= 8 * 3
    assert graphed.stats["backward_replays"] - before["backward_replays"] >= 8 * 3
    assert graphed.stats["busy"] == before["busy"]


def test_replays_read_the_weights_the_optimizer_just_wrote(orc, synth, monkeypatch):
    """The reference loop (zero_grad, forward, nll_loss, backward, Adam step), eight steps through the graphed modules; at
    every step the same pass is also made eagerly FROM THE SAME STATE: log-probabilities bit for bit, gradients to atomics'
    noise -- a replay reads the weights the optimizer wrote a moment ago.  (Whole trajectories cannot be compared: a 1e-6
    perturbation flips ReLU / max-pool gates, which moves gradient entries by per cent, DESIGN.md 2 -- two EAGER runs
    separate to 3e-2 in the activations within two steps.)"""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, kind="facade", seed=302)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    before = dict(graphed.stats)
    losses = []
    for step in range(8):
        buffers = {k: b.clone() for k, b in model.named_buffers()}
        monkeypatch.setattr(graphed, "ENABLED", False)
        want = _train_pass(torch, U, model, x, y, starts, K)
        for k, b in model.named_buffers():
            b.copy_(buffers[k])
        monkeypatch.setattr(graphed, "ENABLED", True)
        opt.zero_grad()
        got = _train_pass(torch, U, model, x, y, starts, K, zero=False)
        _same(torch, got, want)
        opt.step()
        losses.append(float(loss_fn(got[0].reshape(-1, K), y, None)))
    assert graphed.stats["backward_replays"] - before["backward_replays"] >= 8 * 5
    assert graphed.stats["captures"] - before["captures"] == 8
    assert losses[-1] < losses[0] - 0.05


def test_eval_train_eval_and_a_changed_batch_shape(orc, synth, monkeypatch):
    """eval (captured) -> training steps that move weights and running statistics -> eval again through the captured eval
    graphs: equal to an eager eval forward of the trained model (re-derived coefficients, not stale ones); then a batch of
    another shape and the first shape again (a second signature is captured; nothing stale is replayed)."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, B=3, seed=303)

    def ev(inp, st):
        model.eval()
        with torch.no_grad(), U.fps_starts(st):
            return [t.clone() for t in model(inp)]

    def eager(inp, st):
        monkeypatch.setattr(graphed, "ENABLED", False)
        try:
            return ev(inp, st)
        finally:
            monkeypatch.setattr(graphed, "ENABLED", True)
    first = [ev(x, starts) for _ in range(4)]                         # warm-up, capture, replay
    ref = eager(x, starts)
    for got in first:
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(4):
        opt.zero_grad()
        with U.fps_starts(starts):
            pred, _ = model(x)
        loss_fn(pred.contiguous().view(-1, K), y, None).backward()
        opt.step()
    captures = graphed.stats["captures"]
    again = ev(x, starts)
    ref2 = eager(x, starts)
    assert graphed.stats["captures"] == captures                     # the eval graphs of before, replayed
    assert not torch.equal(ref2[0], ref[0])                           # training did move the model
    assert torch.equal(again[0], ref2[0]) and torch.equal(again[1], ref2[1])
    small, st_small = x[:2].contiguous(), [s[:2] for s in starts]
    for _ in range(4):
        got = ev(small, st_small)
    want = eager(small, st_small)
    assert torch.equal(got[0], want[0]) and tuple(got[0].shape)[0] == 2
    back = ev(x, starts)
    assert torch.equal(back[0], ref2[0])


def test_outputs_a_caller_still_holds_are_not_overwritten(orc, synth):
    """A caller that keeps a module's output across the next call finds it intact: the busy instance is left alone and
    another one (captured on demand) serves the call."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, seed=304)
    sa = model.sa1.eval()
    xa, xb = x, torch.flip(x, dims=(2,)).contiguous()
    s0 = [starts[0]]
    with torch.no_grad():
        for _ in range(3):
            with U.fps_starts(s0):
                sa(xa[:, :3, :], xa)
        with U.fps_starts(s0):
            kept = sa(xa[:, :3, :], xa)
        copy = [t.clone() for t in kept]
        captures = graphed.stats["captures"]
        with U.fps_starts(s0):
            other = sa(xb[:, :3, :], xb)
        assert graphed.stats["captures"] == captures + 1
        assert torch.equal(kept[0], copy[0]) and torch.equal(kept[1], copy[1])
        assert not torch.equal(other[1], kept[1])
        del kept, other
        with U.fps_starts(s0):
            again = sa(xa[:, :3, :], xa)
        assert graphed.stats["captures"] == captures + 1             # a free instance again: no further capture
        assert torch.equal(again[1], copy[1])


def test_gradient_accumulation_and_zero_grad_in_place(orc, synth, monkeypatch):
    """Two backward passes without zero_grad (and zero_grad(set_to_none=False), which keeps the adopted buffers): the
    accumulated gradients equal the eager modules' -- the static gradient buffers that autograd adopted as .grad are
    detached before a replay overwrites them."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, seed=305)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run():
        for _ in range(3):
            _train_pass(torch, U, model, x, y, starts, K)             # warm-up / capture
        model.zero_grad(set_to_none=True)
        _train_pass(torch, U, model, x, y, starts, K, zero=False)
        two = _train_pass(torch, U, model, x, y, starts, K, zero=False)
        model.zero_grad(set_to_none=False)
        three = _train_pass(torch, U, model, x, y, starts, K, zero=False)
        return two, three
    monkeypatch.setattr(graphed, "ENABLED", False)
    e2, e3 = run()
    model.load_state_dict(state)
    monkeypatch.setattr(graphed, "ENABLED", True)
    detached = graphed.stats["grad_detached"]
    g2, g3 = run()
    assert graphed.stats["grad_detached"] > detached
    _same(torch, g2, e2)
    _same(torch, g3, e3)
    k = "sa2.mlp_convs.0.weight"
    one = _train_pass(torch, U, model, x, y, starts, K)[2][k]
    assert _close(g2[2][k], 2 * one, 1e-3)                            # two passes on one batch: twice the gradient


def test_momentum_schedule_reaches_a_replayed_module(orc, synth, monkeypatch):
    """The reference loop resets every BatchNorm's momentum per epoch (localfunctions.py:191-195): replays follow it."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, seed=306)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run():
        out = []
        for i in range(6):
            if i == 4:
                for m in model.modules():
                    if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                        m.momentum = 0.5
            out.append(_train_pass(torch, U, model, x, y, starts, K))
        return out
    monkeypatch.setattr(graphed, "ENABLED", False)
    eager = run()
    model.load_state_dict(state)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.momentum = 0.1
    monkeypatch.setattr(graphed, "ENABLED", True)
    got = run()
    for a, b in zip(got, eager):
        _same(torch, a, b)
    assert not torch.equal(eager[5][3]["sa1.mlp_bns.0.running_mean"], eager[3][3]["sa1.mlp_bns.0.running_mean"])


def test_unusual_callers_shared_module_two_forwards_retain_graph(orc, synth, monkeypatch):
    """Callers the loop of the reference is not: ONE module applied to two inputs in a forward (its two calls must not share
    buffers: the second runs on another instance while the first call's backward is pending, and the parameter gradients of
    both arrive), two forwards before either backward, and backward(retain_graph=True) twice.  Every case against the eager
    modules."""
    torch, graphed, U, model, x, y, starts, K = _setup(synth, orc, seed=307)
    sa = model.sa1.train()
    xa = x
    xb = torch.flip(x, dims=(2,)).contiguous()
    s0 = starts[0]

    def siamese():
        sa.zero_grad(set_to_none=True)
        with U.fps_starts([s0, s0]):
            _, fa = sa(xa[:, :3, :], xa)
            _, fb = sa(xb[:, :3, :], xb)
        (fa.square().mean() + 2.0 * fb.mean()).backward()
        return fa.detach().clone(), fb.detach().clone(), {k: p.grad.clone() for k, p in sa.named_parameters()}

    def two_forwards():
        sa.zero_grad(set_to_none=True)
        with U.fps_starts([s0, s0]):
            _, fa = sa(xa[:, :3, :], xa)
            _, fb = sa(xb[:, :3, :], xb)
        la, lb = fa.square().mean(), fb.abs().mean()
        lb.backward()
        gb = {k: p.grad.clone() for k, p in sa.named_parameters()}
        la.backward()
        return fa.detach().clone(), fb.detach().clone(), gb, {k: p.grad.clone() for k, p in sa.named_parameters()}

    def retained():
        sa.zero_grad(set_to_none=True)
        with U.fps_starts([s0]):
            _, fa = sa(xa[:, :3, :], xa)
        loss = fa.square().mean()
        loss.backward(retain_graph=True)
        once = {k: p.grad.clone() for k, p in sa.named_parameters()}
        loss.backward()
        return once, {k: p.grad.clone() for k, p in sa.named_parameters()}

    def grads_close(a, b):
        for k in b:
            if k.endswith(".bias") and "mlp_convs" in k:
                continue
            assert _close(a[k], b[k], 2e-5), k
    state = {k: v.clone() for k, v in sa.state_dict().items()}
    for case in (siamese, two_forwards, retained):
        monkeypatch.setattr(graphed, "ENABLED", False)
        sa.load_state_dict(state)
        want = [case() for _ in range(5)]
        monkeypatch.setattr(graphed, "ENABLED", True)
        sa.load_state_dict(state)
        before = graphed.stats["replays"]
        for i in range(5):
            got = case()
            for a, b in zip(got, want[i]):
                if isinstance(b, dict):
                    grads_close(a, b)
                else:
                    assert torch.equal(a, b)
        assert graphed.stats["replays"] > before, case.__name__
