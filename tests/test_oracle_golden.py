"""Pins the CPU oracle (oracle/pn2_oracle.c + pn2_oracle.py) to outputs of the reference itself
(tests/golden/, written by oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

LEVELS = ((1024, 0.1), (256, 0.2), (64, 0.4), (16, 0.8))
KINDS = ("cube", "facade")


def _levels_xyz(orc, g, blocks):
    xyzs = [np.ascontiguousarray(blocks[:, :, :3])]
    for lv, (npoint, _) in enumerate(LEVELS, start=1):
        xyzs.append(orc.index_points(xyzs[-1], g["fps%d" % lv].astype(np.int64)))
    return xyzs


@pytest.mark.parametrize("kind", KINDS)
def test_fps_and_ball_query_indices_identical(orc, synth, golden, kind):
    g = golden("geometry_" + kind)
    blocks, _, starts, _ = synth.draw_case(int(g["seed"]), int(g["B"]), int(g["N"]), 9, kind)
    cur = np.ascontiguousarray(blocks[:, :, :3])
    for lv, (npoint, radius) in enumerate(LEVELS, start=1):
        assert np.array_equal(starts[lv - 1], g["start%d" % lv])
        fps = orc.farthest_point_sample(cur, npoint, g["start%d" % lv])
        assert np.array_equal(fps, g["fps%d" % lv].astype(np.int64)), "FPS level %d" % lv
        new_xyz = orc.index_points(cur, fps)
        idx = orc.query_ball_point(radius, 32, cur, new_xyz)
        assert np.array_equal(idx, g["ball%d" % lv].astype(np.int64)), "ball query level %d" % lv
        cur = new_xyz


@pytest.mark.parametrize("kind", KINDS)
def test_grouped_tensor(orc, synth, golden, kind):
    g = golden("geometry_" + kind)
    blocks, _, _, _ = synth.draw_case(int(g["seed"]), int(g["B"]), int(g["N"]), 9, kind)
    xyz = np.ascontiguousarray(blocks[:, :, :3])
    new_xyz = orc.index_points(xyz, g["fps1"].astype(np.int64))
    grouped = orc.group_points(xyz, new_xyz, blocks, g["ball1"].astype(np.int64))
    assert grouped.shape == (2, 1024, 32, 12)
    assert np.array_equal(grouped[:, ::41], g["group1_rows"])
    assert grouped.astype(np.float64).sum() == float(g["group1_sum"])
    assert np.abs(grouped.astype(np.float64)).sum() == float(g["group1_abs_sum"])


@pytest.mark.parametrize("kind", KINDS)
def test_square_distance_bit_exact(orc, synth, golden, kind):
    g = golden("geometry_" + kind)
    blocks, _, _, _ = synth.draw_case(int(g["seed"]), int(g["B"]), int(g["N"]), 9, kind)
    xyzs = _levels_xyz(orc, g, blocks)
    d = orc.square_distance(xyzs[1][:, :64], xyzs[0][:, :512])
    assert np.array_equal(d.view(np.uint32), g["sqdist_1024x4096_tile"].view(np.uint32))


@pytest.mark.parametrize("kind", KINDS)
def test_three_nn_and_interpolation(orc, synth, golden, kind):
    g = golden("geometry_" + kind)
    blocks, _, _, _ = synth.draw_case(int(g["seed"]), int(g["B"]), int(g["N"]), 9, kind)
    xyzs = _levels_xyz(orc, g, blocks)
    frs = np.random.RandomState(int(g["nn_feat_seed"]))
    for lv in (3, 2, 1, 0):
        idx, dist, w = orc.three_nn(xyzs[lv], xyzs[lv + 1])
        ok = ~g["nn%d_tie" % lv]                     # reference sort is unstable on exact ties
        assert ok.mean() > 0.99
        assert np.array_equal(idx[ok], g["nn%d_idx" % lv].astype(np.int64)[ok])
        assert np.array_equal(dist[ok], g["nn%d_dist" % lv][ok])
        np.testing.assert_allclose(w[ok], g["nn%d_weight" % lv][ok], rtol=2e-6, atol=1e-7)
        p2 = frs.normal(size=(2, xyzs[lv + 1].shape[1], 16)).astype(np.float32)
        interp = orc.three_interpolate(p2, idx, w)
        np.testing.assert_allclose(interp[ok], g["nn%d_interp" % lv][ok], rtol=1e-5, atol=1e-5)


EVAL_CASES = (("cube", 9, 18), ("facade", 9, 18), ("cube", 6, 18), ("cube", 9, 8), ("facade", 6, 8))


@pytest.mark.parametrize("kind,C,K", EVAL_CASES)
def test_network_eval_logprobs(orc, synth, golden, kind, C, K):
    import torch
    g = golden("model_eval_%s_c%d_k%d" % (kind, C, K))
    blocks, _, starts, _ = synth.draw_case(int(g["seed"]), 1, 4096, C, kind, K)
    for i in range(4):
        assert np.array_equal(starts[i], g["start%d" % (i + 1)])
    net = orc.OracleNet(synth.fill_state_dict(orc.state_shapes(K, C - 6)))
    with torch.no_grad():
        logp, l4 = net.forward(blocks.transpose(0, 2, 1), starts)
    ok = ~g["tie_points"]
    assert np.abs(logp.numpy() - g["logp"])[ok].max() <= 1e-5
    assert np.abs(l4.numpy() - g["l4_points"]).max() <= 1e-5
    if "tap_sa1" in g:
        for name in ("sa1", "sa2", "sa3", "sa4", "fp4", "fp3", "fp2"):
            got = net.taps[name + ".out"].permute(0, 2, 1).numpy()
            assert np.abs(got - g["tap_" + name]).max() <= 1e-5, name
        got = net.taps["fp1.out"].permute(0, 2, 1).numpy()[:, :, ::8]
        assert np.abs(got - g["tap_fp1"])[:, :, ok[0, ::8]].max() <= 1e-5


@pytest.mark.parametrize("kind,C,K", (("cube", 9, 18), ("facade", 6, 8)))
def test_network_train_step(orc, synth, golden, kind, C, K):
    g = golden("model_train_%s_c%d_k%d" % (kind, C, K))
    blocks, labels, starts, cw = synth.draw_case(int(g["seed"]), 2, 4096, C, kind, K)
    assert np.array_equal(cw, g["class_weight"])
    net = orc.OracleNet(synth.fill_state_dict(orc.state_shapes(K, C - 6)), dropout_p=0.0)
    opt = orc.make_adam(net.parameters())
    loss = net.train_step(blocks.transpose(0, 2, 1), labels, starts, opt, cw)
    assert abs(loss - float(g["loss"])) <= 1e-5
    for key in g:
        if key.startswith("grad:"):
            ref = g[key]
            got = net.sd[key[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 1e-5 + 1e-4 * np.abs(ref).max(), key
        elif key.startswith("adam:"):
            # first Adam step moves by lr*sign(g): only defined where |g| is far above its own
            # rounding noise (conv biases under train-mode BN have g == 0 up to noise)
            firm = np.abs(g["grad:" + key[5:]]) > 1e-4
            diff = np.abs(net.sd[key[5:]].detach().numpy() - g[key])
            assert diff[firm].max(initial=0.0) <= 1e-5, key
        elif key.startswith("buf:"):
            assert np.abs(net.sd[key[4:]].numpy() - g[key]).max() <= 1e-5, key


def test_empty_ball_is_an_error(orc):
    xyz = np.zeros((1, 8, 3), np.float32)
    far = np.full((1, 2, 3), 5.0, np.float32)
    with pytest.raises(IndexError):
        orc.query_ball_point(0.1, 4, xyz, far)
    idx = orc.query_ball_point(0.1, 4, xyz, far, allow_empty=True)
    assert (idx == 8).all()          # reference leaves N there and index_points raises (pointnet2_utils.py:59)


def test_ragged_and_padding(orc):
    # one hit only -> tail padded with the first hit; duplicates each take a slot
    xyz = np.array([[[0, 0, 0], [0, 0, 0], [9, 9, 9], [0.05, 0, 0]]], np.float32)
    c = np.array([[[0, 0, 0], [9, 9, 9]]], np.float32)
    idx = orc.query_ball_point(0.1, 4, xyz, c)
    assert idx.tolist() == [[[0, 1, 3, 0], [2, 2, 2, 2]]]
