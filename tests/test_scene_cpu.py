"""SURVEY 8(f) rows on the CPU: the oracle restatements against outputs of the reference's own
functions (tests/golden/scene_*.npz), and the grid-bucketed product classes against both."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from make_golden_scene import make_scene  # noqa: E402  (input construction only; no reference access)
from oracle import scene_oracle as so  # noqa: E402
from khairil_tum_facade_semantic_segmentation_amd import scene  # noqa: E402


def test_add_vote_oracle_matches_reference(golden):
    g = golden("scene_add_vote")
    rs = np.random.RandomState(int(g["seed"]))
    B, N, P, C = 3, 512, 700, 8
    point_idx = rs.randint(0, P, size=(B, N)).astype(np.float64)
    pred = rs.randint(0, C, size=(B, N))
    weight = rs.uniform(0.5, 2.0, size=(B, N))
    weight[rs.rand(B, N) < 0.2] = 0.0
    weight[rs.rand(B, N) < 0.1] = np.inf
    pool = so.add_vote(np.zeros((P, C)), point_idx, pred, weight)
    pool = so.add_vote(pool, point_idx[::-1], pred, weight)
    assert np.array_equal(pool.astype(np.int32), g["pool"])


def _check_tiles(g, out):
    data, lab, wt, idx = out
    assert tuple(data.shape) == tuple(g["shape"])
    assert np.array_equal(idx.astype(np.int32), g["index_room"])
    assert np.array_equal(lab.astype(np.int8), g["label_room"])
    assert np.array_equal(data[0], g["data_first"]) and np.array_equal(data[-1], g["data_last"])
    assert data.sum() == float(g["data_sum"]) and np.abs(data).sum() == float(g["data_abs_sum"])
    assert wt.sum() == float(g["weight_sum"])


def test_tiler_oracle_and_product_match_reference(golden):
    g = golden("scene_tiler")
    xyz, labels, rgb = make_scene(int(g["seed"]), int(g["P"]))
    names = ["red", "blue", "green"]
    np.random.seed(int(g["np_seed"]))
    _check_tiles(g, so.tile_scene(xyz.copy(), labels, rgb, names, g["labelweights"]))
    np.random.seed(int(g["np_seed"]))
    _check_tiles(g, scene.SceneTiler(xyz, labels, rgb, names, g["labelweights"]).tile())


def test_sampler_oracle_and_product_match_reference(golden):
    g = golden("scene_sampler")
    P = int(g["P"])
    rooms = [make_scene(int(g["seeds"][0]), P), make_scene(int(g["seeds"][1]), P // 2, extent=(1.4, 1.2, 2.0))]
    names = ["red", "blue", "green"]

    def check(draw):
        np.random.seed(int(g["np_seed"]))
        feats, labs = zip(*[draw(int(r)) for r in g["room_idxs"]])
        feats, labs = np.stack(feats), np.stack(labs)
        assert np.array_equal(labs.astype(np.int8), g["labels"])
        assert np.array_equal(feats[:, :64], g["feats_first"])
        assert feats.sum() == float(g["feats_sum"]) and np.abs(feats).sum() == float(g["feats_abs_sum"])

    check(lambda r: so.sample_block(rooms[r][0].copy(), rooms[r][1], np.amax(rooms[r][0], axis=0), rooms[r][2], names))
    samplers = [scene.BlockSampler(r[0], r[1], r[2], names) for r in rooms]
    check(lambda r: samplers[r].sample())


def test_grid_index_equals_full_scan():
    rs = np.random.RandomState(3)
    xy = rs.uniform(-3, 5, size=(20000, 2))
    gi = scene.GridIndex(xy, cell=0.3)
    for _ in range(50):
        a, b = np.sort(rs.uniform(-4, 6, size=2)), np.sort(rs.uniform(-4, 6, size=2))
        want = np.where((xy[:, 0] >= a[0]) & (xy[:, 0] <= a[1]) & (xy[:, 1] >= b[0]) & (xy[:, 1] <= b[1]))[0]
        assert np.array_equal(gi.query(a[0], a[1], b[0], b[1]), want)
    assert gi.query(100, 101, 0, 1).size == 0


def test_window_table_enumerates_the_reference_windows():
    """scene.window_table (what the device tiler uploads) = the windows the reference's loop visits: every non-empty
    window of the host tiler (pinned to the reference golden above) selects exactly the points of the table's row."""
    rs = np.random.RandomState(8)
    xyz = rs.uniform(0, 1, size=(20000, 3)) * np.array([2.7, 1.3, 3.0]) + np.array([100.0, -40.0, 2.0])
    cmin, cmax = xyz.min(axis=0), xyz.max(axis=0)
    win, centre = scene.window_table(cmin, cmax)
    gx = int(np.ceil(float(cmax[0] - cmin[0] - 1.0) / 0.5) + 1)
    gy = int(np.ceil(float(cmax[1] - cmin[1] - 1.0) / 0.5) + 1)
    assert win.shape == (gx * gy, 4) and centre.shape == (gx * gy, 2)
    gi = scene.GridIndex(xyz[:, :2], cell=0.25)
    covered = np.zeros(xyz.shape[0], dtype=bool)
    for (x0, x1, y0, y1), (cx, cy) in zip(win, centre):
        assert abs((x1 - x0) - 1.002) < 1e-9 and abs((y1 - y0) - 1.002) < 1e-9           # block_size + 2 * padding
        assert abs(cx - (x0 + 0.001 + 0.5)) < 1e-12 and abs(cy - (y0 + 0.001 + 0.5)) < 1e-12
        sel = gi.query(x0, x1, y0, y1)
        want = np.where((xyz[:, 0] >= x0) & (xyz[:, 0] <= x1) & (xyz[:, 1] >= y0) & (xyz[:, 1] <= y1))[0]
        assert np.array_equal(sel, want)
        covered[sel] = True
    assert covered.all()                                         # the windows tile the whole scene
    assert win[:, 1].max() == cmax[0] + 0.001 and win[:, 3].max() == cmax[1] + 0.001   # the last window is clamped to the scene
