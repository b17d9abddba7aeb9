"""GPU parity of the planned ball query (csrc/pn2_ball_bin.h, pn2_ball_binned.hip, the plan tail of pn2_fps.hip):
identical indices / grouped rows against the reference-generated goldens and the CPU oracle, through both
producers of the plan (the FPS kernel's tail and the stand-alone kernels), over the shapes and edge cases the
self-contained kernels are tested with.  The self-contained entry (pn2_ball_query_group) stays covered by
calling the C ABI directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pn2():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    import khairil_tum_facade_semantic_segmentation_amd as pkg
    from khairil_tum_facade_semantic_segmentation_amd import _lib, ops
    from khairil_tum_facade_semantic_segmentation_amd.models import pointnet2_utils
    _lib.load()
    pkg.ops, pkg.U, pkg.torch, pkg.lib = ops, pointnet2_utils, torch, _lib
    return pkg


def dev(pn2, a):
    return pn2.torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("kind", ("cube", "facade"))
def test_planned_query_matches_golden_both_producers(pn2, orc, synth, golden, kind):
    """SA1 of the reference-generated goldens (B = 2: below the size at which ops picks the planned path by
    itself, so the plan is passed explicitly)."""
    g = golden("geometry_" + kind)
    blocks, _, _, _ = synth.draw_case(int(g["seed"]), 2, 4096, 9, kind)
    xyz = np.ascontiguousarray(blocks[:, :, :3])
    want_idx = g["ball1"].astype(np.int64)
    # producer 1: the tail of the FPS kernel (a plan belongs to the tensors it was built from: the same device tensors
    # go to the query)
    dxyz = dev(pn2, xyz)
    fps, new_xyz, plan = pn2.ops.farthest_point_sample_plan(dxyz, 1024, 0.1, 9, dev(pn2, g["start1"]))
    assert np.array_equal(host(fps), g["fps1"].astype(np.int64))
    want_xyz = orc.index_points(xyz, host(fps))
    assert np.array_equal(host(new_xyz), want_xyz)
    idx, grouped = pn2.ops.ball_query_group(0.1, 32, dxyz, new_xyz, dev(pn2, blocks), plan=plan)
    assert plan.rows_packed
    assert np.array_equal(host(idx), want_idx)
    assert np.array_equal(host(grouped), orc.group_points(xyz, want_xyz, blocks, want_idx))
    assert np.array_equal(host(grouped)[:, ::41], g["group1_rows"])
    # producer 2: the stand-alone kernels, indices only
    plan2 = pn2.ops.ball_plan(0.1, dxyz, new_xyz, None)
    assert np.array_equal(host(pn2.ops.query_ball_point(0.1, 32, dxyz, new_xyz, plan=plan2)), want_idx)
    # a plan for another radius is not used
    other = pn2.ops.ball_plan(0.2, dxyz, new_xyz, None)
    assert np.array_equal(host(pn2.ops.query_ball_point(0.1, 32, dxyz, new_xyz, plan=other)), want_idx)
    pn2.ops.check_errors()


@pytest.mark.parametrize("shape", [
    # B, N, S, K, D, radius, pad_to
    (3, 1500, 300, 32, 9, 0.15, 1),      # N not a multiple of 4, ragged last tile of centroids
    (2, 2048, 512, 16, 5, 0.12, 1),      # 64-word bitmaps, two float4 per row
    (2, 8192, 640, 32, 1, 0.08, 1),      # 256-word bitmaps, one float4 per row
    (2, 4096, 1024, 7, 13, 0.1, 1),      # odd nsample (scalar idx stores), four float4 per row
    (2, 4096, 256, 64, 9, 0.3, 1),       # nsample 64: every ball truncated
    (2, 3000, 500, 32, 6, 0.1, 1),       # 3+D = 9: rows by the separate grouping pass
    (2, 4096, 512, 32, 64, 0.2, 4),      # padded pitch (the MLP's 16-byte rows): separate grouping pass
])
def test_planned_query_shapes(pn2, orc, shape):
    B, N, S, K, D, radius, pad_to = shape
    rs = np.random.RandomState(N + K)
    xyz = rs.uniform(0.0, 1.0, size=(B, N, 3)).astype(np.float32)
    pts = rs.normal(size=(B, N, D)).astype(np.float32)
    start = rs.randint(0, N, size=(B,))
    dxyz, dpts = dev(pn2, xyz), dev(pn2, pts)
    fps, new_xyz, plan = pn2.ops.farthest_point_sample_plan(dxyz, S, radius, D, dev(pn2, start))
    want_fps = orc.farthest_point_sample(xyz, S, start)
    assert np.array_equal(host(fps), want_fps)
    cxyz = orc.index_points(xyz, want_fps)
    want = orc.query_ball_point(radius, K, xyz, cxyz)
    idx, grouped = pn2.ops.ball_query_group(radius, K, dxyz, new_xyz, dpts, pad_to=pad_to, plan=plan)
    assert plan.xyz is dxyz or plan.xyz.data_ptr() == dxyz.data_ptr()
    assert np.array_equal(host(idx), want)
    ref = orc.group_points(xyz, cxyz, pts, want)
    got = host(grouped)
    assert np.array_equal(got[..., :3 + D], ref)
    assert not got[..., 3 + D:].any()
    # stand-alone producer, same answer
    idx2, grouped2 = pn2.ops.ball_query_group(radius, K, dxyz, new_xyz, dpts, pad_to=pad_to,
                                              plan=pn2.ops.ball_plan(radius, dxyz, new_xyz, dpts))
    assert np.array_equal(host(idx2), want)
    assert np.array_equal(host(grouped2), got)
    pn2.ops.check_errors()


def test_planned_query_foreign_centroids_and_empty_balls(pn2, orc):
    """new_xyz need not be a subset of xyz: centroids with a larger norm than any point test every point (the cell
    width was not sized for them), centroids outside the cloud get an empty ball (idx = N, zero rows, error count)."""
    rs = np.random.RandomState(5)
    B, N, S, K, D = 2, 4096, 512, 32, 9
    xyz = rs.uniform(-0.5, 0.5, size=(B, N, 3)).astype(np.float32)
    pts = rs.normal(size=(B, N, D)).astype(np.float32)
    new_xyz = rs.uniform(-0.6, 0.6, size=(B, S, 3)).astype(np.float32)
    new_xyz[:, :8] = np.array([2.0, 2.0, 2.0], np.float32)          # far outside: empty
    want = orc.query_ball_point(0.1, K, xyz, new_xyz, allow_empty=True)
    assert (want == N).any() and (want < N).any()
    dxyz, dnew, dpts = dev(pn2, xyz), dev(pn2, new_xyz), dev(pn2, pts)
    plan = pn2.ops.ball_plan(0.1, dxyz, dnew, dpts)
    idx, grouped = pn2.ops.ball_query_group(0.1, K, dxyz, dnew, dpts, plan=plan)
    assert np.array_equal(host(idx), want)
    got = host(grouped)
    empty = (want == N).all(-1)
    assert not got[empty].any()
    ok = ~empty
    ref = orc.group_points(xyz, new_xyz, pts, np.where(want < N, want, 0))
    assert np.array_equal(got[ok], ref[ok])
    with pytest.raises(IndexError):
        pn2.ops.check_errors()
    pn2.ops._ERR.clear()


def test_self_contained_entry_still_matches(pn2, orc):
    """pn2_ball_query_group (no workspace) keeps its own cell-pruned kernel: same answer as the planned entry."""
    rs = np.random.RandomState(9)
    B, N, S, K, D = 8, 4096, 512, 32, 9
    xyz = rs.uniform(0.0, 1.0, size=(B, N, 3)).astype(np.float32)
    pts = rs.normal(size=(B, N, D)).astype(np.float32)
    pick = np.stack([rs.choice(N, S, replace=False) for _ in range(B)])
    new_xyz = orc.index_points(xyz, pick)
    want = orc.query_ball_point(0.1, K, xyz, new_xyz)
    t = pn2.torch
    dxyz, dnew, dpts = dev(pn2, xyz), dev(pn2, new_xyz), dev(pn2, pts)
    idx = t.empty((B, S, K), dtype=t.int64, device="cuda")
    grouped = t.empty((B, S, K, 3 + D), dtype=t.float32, device="cuda")
    lib = pn2.lib.load()
    rc = lib.pn2_ball_query_group(0.1, K, dxyz.data_ptr(), dnew.data_ptr(), dpts.data_ptr(), B, N, S, D, idx.data_ptr(),
                                  grouped.data_ptr(), 0, None, t.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert np.array_equal(host(idx), want)
    assert np.array_equal(host(grouped), orc.group_points(xyz, new_xyz, pts, want))


def test_plan_argument_checks(pn2):
    lib = pn2.lib.load()
    assert lib.pn2_ball_plan_bytes(4096, 1024, 9) > 0
    assert lib.pn2_ball_plan_bytes(4096, 1024, 9) % 128 == 0
    assert lib.pn2_ball_plan_bytes(8193, 16, 0) == 0
    assert lib.pn2_ball_plan(0.1, None, None, None, 1, 4096, 16, 0, None, None) == -1
    t = pn2.torch
    buf = t.empty(lib.pn2_ball_plan_bytes(2048, 64, 0) + 64, dtype=t.uint8, device="cuda")
    x = t.zeros((1, 2048, 3), device="cuda")
    c = t.zeros((1, 64, 3), device="cuda")
    assert lib.pn2_ball_plan(0.1, x.data_ptr(), c.data_ptr(), None, 1, 2048, 64, 0, buf.data_ptr() + 4, None) == -2   # misaligned
    assert lib.pn2_ball_plan(0.1, x.data_ptr(), c.data_ptr(), None, 1, 9000, 64, 0, buf.data_ptr(), None) == -3       # too many points
