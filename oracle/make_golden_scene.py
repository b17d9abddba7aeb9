#!/usr/bin/env python3
"""Golden vectors for the SURVEY 8(f) rows by RUNNING THE REFERENCE's own functions (build
container only).

`localfunctions.py`, `sem_seg_testing.py` and `sem_seg_training.py` cannot be imported here: their
module-level `import laspy / open3d / h5py` fail (packages absent, no network -- an ordinary
ModuleNotFoundError).  The three functions this script needs use numpy only, so it parses the
reference files with `ast`, compiles exactly those function bodies from where they lie under
/root/reference and calls them; nothing of the reference's text is written to the repository,
only inputs' seeds and the outputs:

  add_vote                             localfunctions.py:339-346
  TestCustomDataset.__getitem__        sem_seg_testing.py:182-254   (sliding-window tiler)
  TrainCustomDataset.__getitem__       sem_seg_training.py:200-259  (training block sampler)

    python oracle/make_golden_scene.py        # writes tests/golden/scene_*.npz
"""
import ast
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from khairil_tum_facade_semantic_segmentation_amd import synth  # noqa: E402


def extract(path, func, cls=None):
    """Compile one function (or one method of `cls`) of a reference file, in a numpy-only namespace."""
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == func)
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns[func]


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def gen_add_vote():
    add_vote = extract(os.path.join(REF, "localfunctions.py"), "add_vote")
    rs = np.random.RandomState(401)
    B, N, P, C = 3, 512, 700, 8
    point_idx = rs.randint(0, P, size=(B, N)).astype(np.float64)       # the reference keeps them as floats
    pred = rs.randint(0, C, size=(B, N))
    weight = rs.uniform(0.5, 2.0, size=(B, N))
    weight[rs.rand(B, N) < 0.2] = 0.0
    weight[rs.rand(B, N) < 0.1] = np.inf
    pool = np.zeros((P, C))
    pool = add_vote(pool, point_idx, pred, weight)
    pool = add_vote(pool, point_idx[::-1], pred, weight)               # a second vote round accumulates
    save("scene_add_vote", seed=np.int64(401), pool=pool.astype(np.int32))


def make_scene(seed, P, extent=(2.3, 1.7, 3.0), num_classes=8):
    """A synthetic 'scene': coordinates offset like raw CRS data, labels, RGB."""
    rs = np.random.RandomState(seed)
    xyz = rs.uniform(0.0, 1.0, size=(P, 3)) * np.asarray(extent) + np.array([10.0, 20.0, 1.0])
    labels = rs.randint(0, num_classes, size=(P,)).astype(np.float64)
    rgb = [rs.randint(0, 256, size=(P,)).astype(np.float64) for _ in range(3)]
    return xyz, labels, rgb


def gen_tiler():
    getitem = extract(os.path.join(REF, "sem_seg_testing.py"), "__getitem__", "TestCustomDataset")
    P, K = 60000, 8
    xyz, labels, rgb = make_scene(402, P)
    lw = np.random.RandomState(5).uniform(0.5, 2.0, size=(K,))
    ds = types.SimpleNamespace(scene_points_list=[xyz.copy()], semantic_labels_list=[labels], block_size=1.0,
                               stride=0.5, padding=0.001, block_points=4096, num_extra_features=3,
                               extra_features_data=[rgb], feature_name=["red", "blue", "green"], labelweights=lw)
    np.random.seed(1234)
    data_room, label_room, sample_weight, index_room = getitem(ds, 0)
    save("scene_tiler", seed=np.int64(402), P=np.int64(P), np_seed=np.int64(1234), labelweights=lw,
         shape=np.array(data_room.shape), index_room=index_room.astype(np.int32),
         label_room=label_room.astype(np.int8), weight_sum=np.float64(sample_weight.sum()),
         data_sum=np.float64(data_room.sum()), data_abs_sum=np.float64(np.abs(data_room).sum()),
         data_first=data_room[0].astype(np.float64), data_last=data_room[-1].astype(np.float64))


def gen_sampler():
    getitem = extract(os.path.join(REF, "sem_seg_training.py"), "__getitem__", "TrainCustomDataset")
    P = 50000
    rooms = [make_scene(403, P), make_scene(404, P // 2, extent=(1.4, 1.2, 2.0))]
    ds = types.SimpleNamespace(
        room_idxs=np.array([0, 1, 1, 0, 1]), room_points=[r[0].copy() for r in rooms], room_labels=[r[1] for r in rooms],
        num_extra_features=3, block_size=1.0, num_point=4096,
        room_coord_max=[np.amax(r[0], axis=0) for r in rooms], feature_name=["red", "blue", "green"],
        extra_features_data=[r[2] for r in rooms], transform=None)
    np.random.seed(4321)
    feats, labs = [], []
    for i in range(len(ds.room_idxs)):
        f, l = getitem(ds, i)
        feats.append(f)
        labs.append(l)
    feats, labs = np.stack(feats), np.stack(labs)
    save("scene_sampler", seeds=np.array([403, 404]), P=np.int64(P), np_seed=np.int64(4321),
         room_idxs=ds.room_idxs, feats_sum=np.float64(feats.sum()), feats_abs_sum=np.float64(np.abs(feats).sum()),
         feats_first=feats[:, :64].astype(np.float64), labels=labs.astype(np.int8))


def gen_labelweights():
    """calculate_labelweights of the training dataset (sem_seg_training.py:264-278): class weights from the rooms' labels."""
    import contextlib
    import io
    fn = extract(os.path.join(REF, "sem_seg_training.py"), "calculate_labelweights", "TrainCustomDataset")
    rs = np.random.RandomState(405)
    K = 8
    rooms = [rs.choice(K, size=n, p=p) for n, p in ((50000, None), (30000, np.array([.3, .2, .1, .1, .1, .1, .05, .05])),
                                                    (8000, np.array([.0, .0, .5, .1, .1, .1, .1, .1])))]
    ds = types.SimpleNamespace(num_classes=K, room_labels=[r.astype(np.float64) for r in rooms])
    with contextlib.redirect_stdout(io.StringIO()):          # the reference prints its intermediate arrays
        w = fn(ds)
    save("scene_labelweights", seed=np.int64(405), sizes=np.array([50000, 30000, 8000]), weights=np.asarray(w, dtype=np.float64))


if __name__ == "__main__":
    gen_labelweights()
    gen_add_vote()
    gen_tiler()
    gen_sampler()
