"""GPU: the less-travelled parts of the reference's operator surface (multi-scale SA, group_all,
returnfps, the [B,C,N] module entry points) against compositions of oracle ops + torch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def U():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils
    return pointnet2_utils


def test_sample_and_group_returnfps_and_group_all(U, orc):
    import torch
    rs = np.random.RandomState(2)
    B, N, D, S, K = 2, 500, 7, 40, 16
    xyz = rs.uniform(-0.5, 0.5, size=(B, N, 3)).astype(np.float32)
    pts = rs.normal(size=(B, N, D)).astype(np.float32)
    start = rs.randint(0, N, size=(B,))
    new_xyz, new_points, grouped_xyz, fps = U.sample_and_group(S, 0.3, K, torch.from_numpy(xyz).cuda(),
                                                               torch.from_numpy(pts).cuda(), returnfps=True,
                                                               start=torch.from_numpy(start).cuda())
    want_fps = orc.farthest_point_sample(xyz, S, start)
    want_xyz = orc.index_points(xyz, want_fps)
    want_idx = orc.query_ball_point(0.3, K, xyz, want_xyz)
    assert np.array_equal(fps.cpu().numpy(), want_fps)
    assert np.array_equal(new_xyz.cpu().numpy(), want_xyz)
    assert np.array_equal(new_points.cpu().numpy(), orc.group_points(xyz, want_xyz, pts, want_idx))
    assert np.array_equal(grouped_xyz.cpu().numpy(), orc.index_points(xyz, want_idx))
    z, g = U.sample_and_group_all(torch.from_numpy(xyz).cuda(), torch.from_numpy(pts).cuda())
    assert z.shape == (B, 1, 3) and float(z.abs().max()) == 0.0
    assert np.array_equal(g.cpu().numpy(), np.concatenate([xyz, pts], -1)[:, None])


def test_modules_channel_first_entry_points(U, orc):
    """PointNetSetAbstraction / Msg / FeaturePropagation forward([B,C,N]) shapes, group_all, and the
    multi-scale channel order [feats, xyz] (reference :248)."""
    import torch
    torch.manual_seed(0)
    B, N, D = 2, 300, 5
    rs = np.random.RandomState(3)
    xyz = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(B, 3, N)).astype(np.float32)).cuda()
    pts = torch.from_numpy(rs.normal(size=(B, D, N)).astype(np.float32)).cuda()
    sa = U.PointNetSetAbstraction(32, 0.4, 8, D + 3, [16, 24], False).cuda().eval()
    with U.fps_starts([np.zeros(B, np.int64)]):
        nx, nf = sa(xyz, pts)
    assert nx.shape == (B, 3, 32) and nf.shape == (B, 24, 32)
    # the same level through oracle ops + torch layers
    x_np, p_np = xyz.permute(0, 2, 1).cpu().numpy(), pts.permute(0, 2, 1).cpu().numpy()
    fps = orc.farthest_point_sample(x_np, 32, np.zeros(B, np.int64))
    cx = orc.index_points(x_np, fps)
    idx = orc.query_ball_point(0.4, 8, x_np, cx)
    g = torch.from_numpy(orc.group_points(x_np, cx, p_np, idx)).cuda().permute(0, 3, 2, 1)      # [B,C,K,S]
    with torch.no_grad():
        for conv, bn in zip(sa.mlp_convs, sa.mlp_bns):
            g = torch.relu(bn(conv(g)))
    want = g.max(2)[0]
    assert float((nf - want).abs().max()) <= 1e-4
    assert np.array_equal(nx.permute(0, 2, 1).cpu().numpy(), cx)

    ga = U.PointNetSetAbstraction(None, None, None, D + 3, [16], True).cuda().eval()
    nx1, nf1 = ga(xyz, pts)
    assert nx1.shape == (B, 3, 1) and nf1.shape == (B, 16, 1)

    msg = U.PointNetSetAbstractionMsg(16, [0.2, 0.5], [4, 8], D, [[8, 8], [8, 12]]).cuda().eval()
    with U.fps_starts([np.zeros(B, np.int64)]):
        nx2, nf2 = msg(xyz, pts)
    assert nx2.shape == (B, 3, 16) and nf2.shape == (B, 20, 16)
    fps = orc.farthest_point_sample(x_np, 16, np.zeros(B, np.int64))
    cx2 = orc.index_points(x_np, fps)
    outs = []
    with torch.no_grad():
        for r, K, convs, bns in zip([0.2, 0.5], [4, 8], msg.conv_blocks, msg.bn_blocks):
            idx = orc.query_ball_point(r, K, x_np, cx2)
            gg = orc.group_points(x_np, cx2, p_np, idx)
            gg = np.concatenate([gg[..., 3:], gg[..., :3]], -1)                                 # [feats, xyz]
            t = torch.from_numpy(gg).cuda().permute(0, 3, 2, 1)
            for conv, bn in zip(convs, bns):
                t = torch.relu(bn(conv(t)))
            outs.append(t.max(2)[0])
    assert float((nf2 - torch.cat(outs, 1)).abs().max()) <= 1e-4

    fp = U.PointNetFeaturePropagation(24 + D, [16]).cuda().eval()
    out = fp(xyz, nx, pts, nf)
    assert out.shape == (B, 16, N)


@pytest.mark.parametrize("N", (512, 4096))
def test_group_all_runs_on_the_hip_stack_and_matches_the_composition(N):
    """PointNetSetAbstraction(group_all=True) (models/pointnet2_utils.py:141-158, :189-190, :200): sample_and_group_all
    + the conv/BN/ReLU stack + a max over ALL N points.  N > 255 pools in two stages on the HIP stack; outputs, the
    running statistics and every gradient against the plain torch composition of the same module."""
    import torch
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils as U
    torch.manual_seed(1)
    B, D = 3, 6
    dev = torch.device("cuda:0")
    xyz = torch.rand(B, 3, N, device=dev)
    pts = torch.randn(B, D, N, device=dev, requires_grad=True)
    sa = U.PointNetSetAbstraction(None, None, None, D + 3, [16, 32], True).to(dev).train()
    ref = U.PointNetSetAbstraction(None, None, None, D + 3, [16, 32], True).to(dev).train()
    ref.load_state_dict(sa.state_dict())
    calls = []
    from khairil_tum_facade_semantic_segmentation_amd import mlp
    real = mlp.mlp_stack
    U.mlp.mlp_stack = lambda *a, **k: (calls.append(a[4] if len(a) > 4 else k.get("pool_k")), real(*a, **k))[1]
    try:
        nx, nf = sa(xyz, pts)
    finally:
        U.mlp.mlp_stack = real
    assert calls == [N], "the stack did not run through the HIP MLP"
    assert nx.shape == (B, 3, 1) and nf.shape == (B, 32, 1)
    g = torch.randn_like(nf)
    (nf * g).sum().backward()
    # torch composition: [xyz, points] -> conv/bn/relu x2 -> max over the N points
    pts2 = pts.detach().clone().requires_grad_(True)
    x = torch.cat([xyz, pts2], dim=1).unsqueeze(-1)                           # [B, C, K = N samples, S = 1 centroid]
    for conv, bn in zip(ref.mlp_convs, ref.mlp_bns):
        x = torch.relu(bn(conv(x)))
    want = x.max(2)[0]
    (want * g).sum().backward()
    assert float((nf - want).abs().max()) <= 1e-4
    assert float((pts.grad - pts2.grad).abs().max()) <= 1e-4 * float(pts2.grad.abs().max()) + 1e-6
    for (k, p), (_, q) in zip(sa.named_parameters(), ref.named_parameters()):
        if "mlp_convs" in k and k.endswith("bias"):
            continue
        assert float((p.grad - q.grad).abs().max()) <= 2e-3 * float(q.grad.abs().max()) + 1e-6, k
    for (k, a), (_, b) in zip(sa.named_buffers(), ref.named_buffers()):
        assert float((a.double() - b.double()).abs().max()) <= 1e-4 * (float(b.double().abs().max()) + 1.0), k
