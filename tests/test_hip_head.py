"""GPU numerics of the segmentation-head tail and the loss (csrc/pn2_head.hip through the C ABI)
against a plain PyTorch fp32/fp64 CPU reference of the same ops: log_softmax(conv2(x)) and
F.nll_loss(pred, target, weight) -- reference models/pointnet2_sem_seg.py:37-38,48."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from khairil_tum_facade_semantic_segmentation_amd import _lib, head
    _lib.load()
    return torch, head


@pytest.mark.parametrize("M,K,C", [(4096, 128, 13), (1000, 128, 8), (77, 64, 17), (16 * 4096, 128, 13), (3, 32, 32), (200, 128, 32), (65, 128, 18)])
def test_head_logits_forward_backward(env, M, K, C):
    torch, head = env
    g = torch.Generator().manual_seed(M + K + C)
    y = torch.randn(M, K, generator=g)
    w = torch.randn(C, K, 1, generator=g) * 0.2
    b = torch.randn(C, generator=g) * 0.1
    up = torch.randn(M, C, generator=g)

    yr, wr, br = (t.double().requires_grad_() for t in (y, w, b))
    ref = torch.log_softmax(torch.nn.functional.conv1d(yr.t().unsqueeze(0), wr, br)[0].t(), dim=1)
    (ref * up.double()).sum().backward()

    yd, wd, bd = (t.cuda().requires_grad_() for t in (y, w, b))
    out = head.head_logits(yd, wd, bd)
    assert out.shape == (M, C)
    (out * up.cuda()).sum().backward()
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=0, atol=2e-5)   # log-probs (tolerance: north_star 1e-3)
    scale = lambda t: float(t.abs().max()) + 1e-12
    for name, got, want in (("gy", yd.grad, yr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        err = float((got.cpu().double() - want).abs().max()) / scale(want)
        assert err < 2e-5, (name, err)


def test_head_logits_no_input_grad(env):
    torch, head = env
    y = torch.randn(300, 128).cuda()
    conv = torch.nn.Conv1d(128, 13, 1).cuda()
    out = head.head_logits(y, conv.weight, conv.bias)
    out[:, 3].sum().backward()
    ref = torch.log_softmax(conv(y.t().unsqueeze(0))[0].t(), dim=1)
    assert conv.weight.grad.shape == conv.weight.shape
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("M,C,weighted", [(65536, 13, True), (1000, 8, False), (5, 3, True)])
def test_nll_loss(env, M, C, weighted):
    torch, head = env
    g = torch.Generator().manual_seed(M)
    logp = torch.log_softmax(torch.randn(M, C, generator=g), dim=1)
    t = torch.randint(0, C, (M,), generator=g)
    if M > 100:
        t[::17] = -100                                    # ignore_index rows are skipped
    wt = (torch.rand(C, generator=g) + 0.5) if weighted else None

    lr = logp.double().requires_grad_()
    ref = torch.nn.functional.nll_loss(lr, t, weight=None if wt is None else wt.double())
    (ref * 1.7).backward()

    ld = logp.cuda().requires_grad_()
    out = head.nll_loss(ld, t.cuda(), None if wt is None else wt.cuda())
    (out * 1.7).backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 1e-6 * max(1.0, abs(float(ref.detach())))
    torch.testing.assert_close(ld.grad.cpu().double(), lr.grad, rtol=1e-5, atol=1e-9)


def test_nll_loss_single_launch_is_repeatable(env):
    """The loss is finished by whichever workgroup ends last (pn2_nll_loss_ticketed): launch after launch on the same
    ticket word, eagerly and replayed from a graph, gives the bits of the two-launch form (fixed summation order)."""
    torch, head = env
    from khairil_tum_facade_semantic_segmentation_amd import _lib
    from khairil_tum_facade_semantic_segmentation_amd.ops import _ptr, _stream
    lib = _lib.load()
    M, C = 300000, 13
    g = torch.Generator().manual_seed(1)
    logp = torch.log_softmax(torch.randn(M, C, generator=g), dim=1).cuda()
    t = torch.randint(0, C, (M,), generator=g).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    P = lib.pn2_nll_loss_partials(M)
    part = torch.empty((P, 2), dtype=torch.float64, device="cuda")
    two = torch.empty(2, device="cuda")
    rc = lib.pn2_nll_loss(_ptr(logp), _ptr(t), _ptr(w), M, C, -100, _ptr(part), _ptr(two), two.data_ptr() + 4, None,
                          _stream(logp.device))
    assert rc == 0
    want = two.clone()
    for _ in range(20):
        out = head.nll_loss(logp, t, w)
        assert float(out) == float(want[0])
    head.ensure_ticket_words(logp.device)
    static = torch.zeros((), device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(3):
            static.copy_(head.nll_loss(logp, t, w))
    for _ in range(3):
        static.zero_()
        graph.replay()
        assert float(static) == float(want[0])
    assert all(int(v) == 0 for v in head._tickets.values())


def test_nll_loss_bad_target_is_reported(env):
    torch, head = env
    from khairil_tum_facade_semantic_segmentation_amd import ops
    logp = torch.log_softmax(torch.randn(64, 5), dim=1).cuda()
    t = torch.randint(0, 5, (64,))
    t[7] = 9
    head.nll_loss(logp, t.cuda())
    with pytest.raises(IndexError):
        ops.check_errors()


def test_head_rejects_cpu_tensors(env):
    torch, head = env
    with pytest.raises(RuntimeError):
        head.head_logits(torch.randn(8, 128), torch.randn(13, 128), None)
    with pytest.raises(RuntimeError):
        head.nll_loss(torch.randn(8, 13), torch.zeros(8, dtype=torch.int64))


@pytest.mark.parametrize("M,K,C,p", [(4096, 128, 13, 0.5), (1000, 128, 8, 0.3), (77, 64, 17, 0.5), (65, 128, 18, 0.9)])
def test_head_logits_fused_dropout(env, M, K, C, p):
    """Dropout in front of conv2 applied inside the kernels: forward and backward must use the SAME keep-mask (the one
    pn2_dropout_mask reports for the seed), i.e. equal torch ops on y * mask / (1 - p)."""
    torch, head = env
    g = torch.Generator().manual_seed(M + K + C)
    y = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(C, K, 1, generator=g) * 0.2).cuda().requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True)
    seed = torch.tensor([0x1234567 + M], dtype=torch.int64, device="cuda")
    mask = head.dropout_mask(seed, p, M, K)
    rate = float(mask.float().mean())
    assert abs(rate - (1.0 - p)) < 4.0 * (p * (1 - p) / (M * K)) ** 0.5 + 1e-3, rate
    other = head.dropout_mask(seed + 1, p, M, K)
    assert float((mask != other).float().mean()) > 0.5 * 2 * p * (1 - p)

    out = head.head_logits(y, w, b, drop_p=p, seed=seed)
    yr, wr, br = (t.detach().clone().requires_grad_(True) for t in (y, w, b))
    ref = torch.log_softmax((yr * mask / (1.0 - p)) @ wr.reshape(C, K).t() + br, dim=1)
    assert float((out - ref).detach().abs().max()) <= 2e-5 * (float(ref.detach().abs().max()) + 1.0)
    go = torch.randn(M, C, generator=g).cuda()
    out.backward(go)
    ref.backward(go)
    for a, r, what in ((y.grad, yr.grad, "gy"), (w.grad, wr.grad, "dw"), (b.grad, br.grad, "db")):
        s = float(r.abs().max()) + 1e-6
        assert float((a - r).abs().max()) <= 2e-4 * s + 1e-6, what
    # a dropped element receives no gradient
    assert float(y.grad[~mask].abs().max() if (~mask).any() else 0.0) == 0.0


@pytest.mark.parametrize("M,K,C", [(4096, 128, 13), (77, 64, 17)])
def test_head_dropout_counted_seed(env, M, K, C):
    """Without an explicit seed the forward hashes its seed from a device counter (pn2_head_logits_dropout_counted): the
    backward uses the seed the forward reports, every call draws a new mask, the call counter advances by one per call, and
    the result equals the explicit-seed form run with the reported seed."""
    torch, head = env
    if not head._COUNTED_DROPOUT:
        pytest.skip("PN2_COUNTED_DROPOUT=0")
    g = torch.Generator().manual_seed(M + C)
    y = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(C, K, 1, generator=g) * 0.2).cuda().requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True)
    state = head._dropout_state(y.device)
    calls0 = int(state[1])
    outs, seeds = [], []
    for _ in range(3):
        out = head.head_logits(y, w, b, drop_p=0.5)
        seeds.append(out.grad_fn.seed.clone())
        outs.append(out)
    assert int(state[1]) == calls0 + 3 and int(state[2]) == 0
    assert len({int(s) for s in seeds}) == 3
    assert not torch.equal(outs[0], outs[1])
    # the same numbers as the explicit-seed form with the seed the forward reported; the backward masks with it
    y2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (y, w, b))
    ref = head.head_logits(y2, w2, b2, drop_p=0.5, seed=seeds[2])
    assert torch.equal(ref, outs[2])
    go = torch.randn(M, C, generator=g).cuda()
    outs[2].backward(go)
    ref.backward(go)
    assert torch.equal(y.grad, y2.grad) and torch.equal(w.grad, w2.grad) and torch.equal(b.grad, b2.grad)
    mask = head.dropout_mask(seeds[2], 0.5, M, K)
    assert float(y.grad[~mask].abs().max()) == 0.0
