"""CPU ORACLE for the SURVEY 8(f) rows -- test infrastructure, NOT the product path.

Restates, step for step and with the same numpy RNG call sequence, three host routines of the
reference.  Pinned against outputs of the reference's own functions (tests/golden/scene_*.npz,
made by oracle/make_golden_scene.py); see tests/test_scene_cpu.py."""
import numpy as np


def add_vote(vote_label_pool, point_idx, pred_label, weight):
    """localfunctions.py:339-346."""
    B, N = pred_label.shape
    for b in range(B):
        for n in range(N):
            w = weight[b, n]
            if w != 0 and not np.isinf(w):
                vote_label_pool[int(point_idx[b, n]), int(pred_label[b, n])] += 1
    return vote_label_pool


def tile_scene(points_all, labels, extra, feature_name, labelweights, block_points=4096, block_size=1.0,
               stride=0.5, padding=0.001):
    """Sliding-window tiling of one scene: TestCustomDataset.__getitem__, sem_seg_testing.py:182-254.
    points_all [P,>=3] raw coordinates, extra = list of per-point feature arrays.  Uses the global
    numpy RNG exactly like the reference (choice, then shuffle, per window)."""
    points = points_all[:, :3]
    cmin, cmax = np.amin(points, axis=0)[:3], np.amax(points, axis=0)[:3]            # :186
    gx = int(np.ceil(float(cmax[0] - cmin[0] - block_size) / stride) + 1)              # :187
    gy = int(np.ceil(float(cmax[1] - cmin[1] - block_size) / stride) + 1)              # :188
    datas, labs, wts, idxs = [], [], [], []
    for iy in range(gy):                                                              # :194
        for ix in range(gx):
            sx = cmin[0] + ix * stride
            ex = min(sx + block_size, cmax[0])
            sx = ex - block_size
            sy = cmin[1] + iy * stride
            ey = min(sy + block_size, cmax[1])
            sy = ey - block_size
            sel = np.where((points[:, 0] >= sx - padding) & (points[:, 0] <= ex + padding) &
                           (points[:, 1] >= sy - padding) & (points[:, 1] <= ey + padding))[0]    # :202-203
            if sel.size == 0:
                continue
            nb = int(np.ceil(sel.size / block_points))
            size = int(nb * block_points)
            replace = False if (size - sel.size <= sel.size) else True                            # :209
            rep = np.random.choice(sel, size - sel.size, replace=replace)
            sel = np.concatenate((sel, rep))
            np.random.shuffle(sel)                                                                # :212
            batch = points[sel, :]                     # fancy indexing: a copy, as in the reference
            norm = np.zeros((size, 3))
            norm[:, 0] = batch[:, 0] / cmax[0]                                                     # :217-219
            norm[:, 1] = batch[:, 1] / cmax[1]
            norm[:, 2] = batch[:, 2] / cmax[2]
            batch[:, 0] = batch[:, 0] - (sx + block_size / 2.0)                                    # :220-221
            batch[:, 1] = batch[:, 1] - (sy + block_size / 2.0)
            batch = np.concatenate((batch, norm), axis=1)
            lab = labels[sel].astype(int)
            wt = labelweights[lab]
            if len(extra) > 0:                                                                    # :227-239
                ex_cols = np.zeros((size, len(extra)))
                for i, name in enumerate(feature_name):
                    f = extra[i][sel]
                    if name in ("red", "blue", "green"):
                        f = f / 255
                    ex_cols[:, i] = np.array(f)
                batch = np.concatenate((batch, ex_cols), axis=1)
            datas.append(batch)
            labs.append(lab)
            wts.append(wt)
            idxs.append(sel)
    data = np.vstack(datas).reshape((-1, block_points, datas[0].shape[1]))            # :249-252
    return (data, np.hstack(labs).reshape((-1, block_points)), np.hstack(wts).reshape((-1, block_points)),
            np.hstack(idxs).reshape((-1, block_points)))


def sample_block(points, labels, coord_max, extra, feature_name, num_point=4096, block_size=1.0):
    """One training block: TrainCustomDataset.__getitem__, sem_seg_training.py:200-259 (transform=None).
    points [P,3] raw coordinates of the room.  Global numpy RNG, same call order as the reference."""
    P = points.shape[0]
    while True:                                                                       # :207-216
        center = points[np.random.choice(P)][:3]
        bmin = center - [block_size / 2.0, block_size / 2.0, 0]
        bmax = center + [block_size / 2.0, block_size / 2.0, 0]
        sel = np.where((points[:, 0] >= bmin[0]) & (points[:, 0] <= bmax[0]) &
                       (points[:, 1] >= bmin[1]) & (points[:, 1] <= bmax[1]))[0]
        if sel.size > 1024:
            break
    if sel.size >= num_point:                                                         # :218-221
        chosen = np.random.choice(sel, num_point, replace=False)
    else:
        chosen = np.random.choice(sel, num_point, replace=True)
    picked = points[chosen, :]
    cur = np.zeros((num_point, 6))
    cur[:, 3] = picked[:, 0] / coord_max[0]                                           # :226-228
    cur[:, 4] = picked[:, 1] / coord_max[1]
    cur[:, 5] = picked[:, 2] / coord_max[2]
    picked[:, 0] = picked[:, 0] - center[0]                                           # :229-231
    picked[:, 1] = picked[:, 1] - center[1]
    cur[:, 0:3] = picked
    out = np.zeros((num_point, 6 + len(extra)))
    out[:, :6] = cur
    for i, name in enumerate(feature_name):                                           # :237-252
        f = extra[i][chosen]
        if name in ("red", "blue", "green"):
            f = f / 255
        out[:, 6 + i] = f
    return out, labels[chosen]
