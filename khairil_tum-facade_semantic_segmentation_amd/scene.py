"""Host/device pieces either side of the hot path: the training block sampler, the
whole-scene sliding-window tiler and the vote aggregation of whole-scene inference.

The reference does an `np.where` over the ENTIRE scene for every block / window
(sem_seg_training.py:211, sem_seg_testing.py:202) and adds votes in a Python double loop
(localfunctions.py:339-346); at GPU speeds those, not the network, bound end-to-end throughput.
Here the scene's points are bucketed once into a 2-D grid so a window touches only the cells it
overlaps, and votes are scattered on the device straight from the network's log-probabilities.
Block / window semantics and the numpy RNG call sequence are those of the reference, so the same
seed yields the same blocks (tests/test_scene_cpu.py, against outputs of the reference's code)."""
import os

import numpy as np


class GridIndex:
    """Points of one scene bucketed by (x, y) cell; query(xmin, xmax, ymin, ymax) returns the
    indices of the points inside the closed window in ascending order -- what
    `np.where(mask)[0]` yields in the reference -- touching only the overlapped cells."""

    def __init__(self, xy, cell=0.25):
        xy = np.asarray(xy, dtype=np.float64)
        self.xy = xy
        self.cell = float(cell)
        self.origin = xy.min(axis=0)
        ij = np.floor((xy - self.origin) / self.cell).astype(np.int64)
        self.nx, self.ny = int(ij[:, 0].max()) + 1, int(ij[:, 1].max()) + 1
        key = ij[:, 1] * self.nx + ij[:, 0]
        order = np.argsort(key, kind="stable")                 # stable: ascending index inside a cell
        self.order = order
        self.start = np.searchsorted(key[order], np.arange(self.nx * self.ny + 1), side="left")

    def query(self, xmin, xmax, ymin, ymax):
        c = self.cell
        i0 = max(int(np.floor((xmin - self.origin[0]) / c)), 0)
        i1 = min(int(np.floor((xmax - self.origin[0]) / c)), self.nx - 1)
        j0 = max(int(np.floor((ymin - self.origin[1]) / c)), 0)
        j1 = min(int(np.floor((ymax - self.origin[1]) / c)), self.ny - 1)
        if i1 < i0 or j1 < j0:
            return np.empty(0, dtype=np.int64)
        parts = [self.order[self.start[j * self.nx + i0]:self.start[j * self.nx + i1 + 1]] for j in range(j0, j1 + 1)]
        cand = np.concatenate(parts)
        p = self.xy[cand]
        keep = (p[:, 0] >= xmin) & (p[:, 0] <= xmax) & (p[:, 1] >= ymin) & (p[:, 1] <= ymax)
        return np.sort(cand[keep])


def _extra_columns(extra, feature_name, sel):
    cols = np.zeros((sel.size, len(extra)))
    for i, name in enumerate(feature_name):
        f = extra[i][sel]
        cols[:, i] = f / 255 if name in ("red", "blue", "green") else f      # sem_seg_testing.py:233-234
    return cols


class SceneTiler:
    """Sliding-window blocks of a whole scene (TestCustomDataset.__getitem__,
    sem_seg_testing.py:182-254): windows of block_size at `stride`, clamped to the scene, padded by
    `padding`; each window's points are topped up to a multiple of block_points and shuffled."""

    def __init__(self, points, labels, extra=(), feature_name=(), labelweights=None, block_points=4096,
                 block_size=1.0, stride=0.5, padding=0.001):
        self.points = np.asarray(points, dtype=np.float64)[:, :3]
        self.labels = np.asarray(labels)
        self.extra = list(extra)
        self.feature_name = list(feature_name)
        self.labelweights = np.ones(int(self.labels.max()) + 1) if labelweights is None else np.asarray(labelweights)
        self.block_points, self.block_size, self.stride, self.padding = block_points, block_size, stride, padding
        self.index = GridIndex(self.points[:, :2], cell=block_size / 4.0)

    def tile(self):
        """-> data [nblocks, block_points, 6+extra], labels, sample weights, point indices."""
        pts, bs, st, pad, bp = self.points, self.block_size, self.stride, self.padding, self.block_points
        cmin, cmax = pts.min(axis=0), pts.max(axis=0)
        gx = int(np.ceil(float(cmax[0] - cmin[0] - bs) / st) + 1)
        gy = int(np.ceil(float(cmax[1] - cmin[1] - bs) / st) + 1)
        datas, labs, wts, idxs = [], [], [], []
        for iy in range(gy):
            for ix in range(gx):
                ex = min(cmin[0] + ix * st + bs, cmax[0])
                sx = ex - bs
                ey = min(cmin[1] + iy * st + bs, cmax[1])
                sy = ey - bs
                sel = self.index.query(sx - pad, ex + pad, sy - pad, ey + pad)
                if sel.size == 0:
                    continue
                size = int(np.ceil(sel.size / bp)) * bp
                fill = size - sel.size
                rep = np.random.choice(sel, fill, replace=fill > sel.size)      # :209-210
                sel = np.concatenate((sel, rep))
                np.random.shuffle(sel)                                         # :212
                xyz = pts[sel]
                block = np.empty((size, 6 + len(self.extra)))
                block[:, 0] = xyz[:, 0] - (sx + bs / 2.0)
                block[:, 1] = xyz[:, 1] - (sy + bs / 2.0)
                block[:, 2] = xyz[:, 2]
                block[:, 3:6] = xyz / cmax
                if self.extra:
                    block[:, 6:] = _extra_columns(self.extra, self.feature_name, sel)
                lab = self.labels[sel].astype(int)
                datas.append(block)
                labs.append(lab)
                wts.append(self.labelweights[lab])
                idxs.append(sel)
        data = np.concatenate(datas).reshape(-1, bp, 6 + len(self.extra))
        return (data, np.concatenate(labs).reshape(-1, bp), np.concatenate(wts).reshape(-1, bp),
                np.concatenate(idxs).reshape(-1, bp))


class BlockSampler:
    """Random training blocks of one room (TrainCustomDataset.__getitem__, sem_seg_training.py:200-259):
    a block_size column around a random point, re-drawn until it holds > 1024 points, num_point of
    them sampled (with replacement only if there are fewer); x, y centred, xyz / room_max appended."""

    def __init__(self, points, labels, extra=(), feature_name=(), num_point=4096, block_size=1.0):
        self.points = np.asarray(points, dtype=np.float64)[:, :3]
        self.labels = np.asarray(labels)
        self.extra = list(extra)
        self.feature_name = list(feature_name)
        self.num_point, self.block_size = num_point, block_size
        self.coord_max = self.points.max(axis=0)
        self.index = GridIndex(self.points[:, :2], cell=block_size / 4.0)

    def sample(self):
        pts, half = self.points, self.block_size / 2.0
        while True:
            center = pts[np.random.choice(pts.shape[0])]
            sel = self.index.query(center[0] - half, center[0] + half, center[1] - half, center[1] + half)
            if sel.size > 1024:
                break
        chosen = np.random.choice(sel, self.num_point, replace=sel.size < self.num_point)
        xyz = pts[chosen]
        out = np.empty((self.num_point, 6 + len(self.extra)))
        out[:, 0] = xyz[:, 0] - center[0]
        out[:, 1] = xyz[:, 1] - center[1]
        out[:, 2] = xyz[:, 2]
        out[:, 3:6] = xyz / self.coord_max
        if self.extra:
            out[:, 6:] = _extra_columns(self.extra, self.feature_name, chosen)
        return out, self.labels[chosen]


class DeviceBlockSampler:
    """BlockSampler with the scene resident on the device: the grid index is built once on the
    host (GridIndex) and uploaded; sample(B, seed) draws B training blocks with one kernel launch
    (csrc/pn2_sampler.hip; same block semantics as TrainCustomDataset.__getitem__, randomness from a counter-based
    hash of the seed), sample_exact() replays the reference's numpy random stream -- host-drawn centre indices and
    choice() positions -- and returns float64 features that equal the reference's bit for bit."""

    def __init__(self, points, labels, extra=(), feature_name=(), num_point=4096, block_size=1.0, device="cuda"):
        import torch
        self.torch = torch
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :3])
        self.P = pts.shape[0]
        self.num_point, self.block_size = int(num_point), float(block_size)
        self.coord_max = pts.max(axis=0)
        gi = GridIndex(pts[:, :2], cell=block_size / 4.0)
        self.grid = gi
        dev = torch.device(device)
        self.dev = dev
        self.xyz = torch.from_numpy(pts).to(dev)
        self.order = torch.from_numpy(gi.order.astype(np.int32)).to(dev)
        self.cell_start = torch.from_numpy(gi.start.astype(np.int32)).to(dev)
        self.labels = torch.from_numpy(np.asarray(labels).astype(np.int64)).to(dev)
        self.feature_name = list(feature_name)
        self.extra_raw = [torch.from_numpy(np.ascontiguousarray(np.asarray(e))).to(dev) for e in extra]
        cols = []
        for e, name in zip(extra, self.feature_name):
            e = np.asarray(e, dtype=np.float64)
            cols.append((e / 255 if name in ("red", "blue", "green") else e).astype(np.float32))      # :241-243
        self.E = len(cols)
        self.extra = torch.from_numpy(np.stack(cols)).to(dev) if cols else None

    def sample(self, B, seed, want_indices=False):
        """-> feats [B, num_point, 6+E] float32, labels [B, num_point] int64, info [B, 4] int32 (centre index,
        points in the window, attempts, gave-up flag) [, indices [B, num_point] int32]"""
        from . import _lib
        import ctypes
        torch = self.torch
        lib = _lib.load()
        F = 6 + self.E
        feats = torch.empty((B, self.num_point, F), dtype=torch.float32, device=self.dev)
        labs = torch.empty((B, self.num_point), dtype=torch.int64, device=self.dev)
        info = torch.empty((B, 4), dtype=torch.int32, device=self.dev)
        sel = torch.empty((B, self.num_point), dtype=torch.int32, device=self.dev) if want_indices else None
        cm = (ctypes.c_double * 3)(*[float(v) for v in self.coord_max])
        gi = self.grid
        with torch.cuda.device(self.dev):
            rc = lib.pn2_sample_blocks(self.xyz.data_ptr(), self.order.data_ptr(), self.cell_start.data_ptr(),
                                       None if self.extra is None else self.extra.data_ptr(), self.labels.data_ptr(),
                                       float(gi.origin[0]), float(gi.origin[1]), float(gi.cell), gi.nx, gi.ny, self.P, self.E,
                                       self.block_size, cm, self.num_point, 1024, int(seed) & (2 ** 64 - 1), B, feats.data_ptr(),
                                       labs.data_ptr(), info.data_ptr(), None if sel is None else sel.data_ptr(),
                                       torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "pn2_sample_blocks")
        return (feats, labs, info, sel) if want_indices else (feats, labs, info)

    def sample_exact(self, rng=np.random):
        """One block with the reference's own random stream: rng.choice(P) per attempt (:207), then rng.choice over the
        window's points (:219-221).  float64 features [num_point, 6+E] and labels, bit-identical to the reference."""
        torch = self.torch
        half = self.block_size / 2.0
        x, y = self.xyz[:, 0], self.xyz[:, 1]
        while True:
            ci = int(rng.choice(self.P))
            c = self.xyz[ci]
            cx, cy = float(c[0]), float(c[1])
            mask = (x >= cx - half) & (x <= cx + half) & (y >= cy - half) & (y <= cy + half)
            idxs = torch.nonzero(mask).squeeze(1)                      # ascending, like np.where
            if idxs.numel() > 1024:
                break
        cnt = int(idxs.numel())
        pos = rng.choice(cnt, self.num_point, replace=cnt < self.num_point)
        sel = idxs[torch.from_numpy(np.asarray(pos, dtype=np.int64)).to(self.dev)]
        sp = self.xyz[sel]
        out = torch.empty((self.num_point, 6 + self.E), dtype=torch.float64, device=self.dev)
        cm = torch.from_numpy(self.coord_max).to(self.dev)
        out[:, 3:6] = sp / cm
        out[:, 0] = sp[:, 0] - c[0]
        out[:, 1] = sp[:, 1] - c[1]
        out[:, 2] = sp[:, 2]
        # a DEVICE divisor: torch turns `tensor / python_scalar` into a multiplication by the reciprocal on the GPU,
        # which is not the correctly rounded quotient the reference's numpy division yields
        c255 = torch.full((1,), 255.0, dtype=torch.float64, device=self.dev)
        for k, (raw, name) in enumerate(zip(self.extra_raw, self.feature_name)):
            f = raw[sel].to(torch.float64)
            out[:, 6 + k] = f / c255 if name in ("red", "blue", "green") else f
        return out.cpu().numpy(), self.labels[sel].cpu().numpy()


def window_table(cmin, cmax, block_size=1.0, stride=0.5, padding=0.001):
    """The sliding-window grid of TestCustomDataset.__getitem__ (sem_seg_testing.py:187-201) in the reference's order (y
    outer, x inner) and float64 arithmetic: -> (windows [W,4] = xmin, xmax, ymin, ymax of the closed, padded window;
    centres [W,2] = the point the block's x / y are centred on, :220-221)."""
    bs, st, pad = float(block_size), float(stride), float(padding)
    gx = int(np.ceil(float(cmax[0] - cmin[0] - bs) / st) + 1)
    gy = int(np.ceil(float(cmax[1] - cmin[1] - bs) / st) + 1)
    win, centre = [], []
    for iy in range(gy):
        for ix in range(gx):
            ex = min(cmin[0] + ix * st + bs, cmax[0])
            sx = ex - bs
            ey = min(cmin[1] + iy * st + bs, cmax[1])
            sy = ey - bs
            win.append((sx - pad, ex + pad, sy - pad, ey + pad))
            centre.append((sx + bs / 2.0, sy + bs / 2.0))
    return np.asarray(win, dtype=np.float64).reshape(-1, 4), np.asarray(centre, dtype=np.float64).reshape(-1, 2)


class MultiRoomSampler:
    """Several DeviceBlockSamplers behind ONE launch per batch (pn2_sample_blocks_multi): the reference's loader mixes rooms
    inside a batch (room_idxs replicated by point share and shuffled, sem_seg_training.py:184-193), and a launch per
    contributing room costs 0.1 ms each in front of every step (measured: 3.41 against 2.93 ms per step with four rooms).
    The per-room descriptors (device pointers, grid, room maxima) live in a device table built once."""

    def __init__(self, samplers):
        import struct
        import torch
        self.torch = torch
        self.samplers = list(samplers)
        s0 = self.samplers[0]
        if any((s.E, s.num_point, s.block_size, s.dev) != (s0.E, s0.num_point, s0.block_size, s0.dev) for s in self.samplers):
            raise ValueError("rooms must share the extra-feature count, num_point, block_size and device")
        self.dev, self.E, self.num_point, self.block_size = s0.dev, s0.E, s0.num_point, s0.block_size
        self.sizes = [s.P for s in self.samplers]
        blob = b""
        for s in self.samplers:
            gi = s.grid
            blob += struct.pack("5Q6d4i", s.xyz.data_ptr(), s.order.data_ptr(), s.cell_start.data_ptr(),
                                0 if s.extra is None else s.extra.data_ptr(), s.labels.data_ptr(), float(gi.origin[0]),
                                float(gi.origin[1]), float(gi.cell), float(s.coord_max[0]), float(s.coord_max[1]),
                                float(s.coord_max[2]), gi.nx, gi.ny, s.P, 0)
        self.table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.dev)

    def sample(self, room_of_block, seed, want_indices=False):
        """room_of_block: [B] room indices (array / tensor) -> feats [B, num_point, 6+E], labels [B, num_point], info [B, 4]
        (centre index in its room, points in the window, attempts, gave-up flag) [, indices into the room [B, num_point]]"""
        from . import _lib
        torch = self.torch
        lib = _lib.load()
        ids = np.ascontiguousarray(np.asarray(room_of_block.cpu() if hasattr(room_of_block, "cpu") else room_of_block, dtype=np.int32))
        B = int(ids.shape[0])
        on_host = B <= 256 and len(self.samplers) <= 256       # the ids then travel in the launch's arguments: no upload
        rooms = None if on_host else torch.from_numpy(ids).to(self.dev)
        feats = torch.empty((B, self.num_point, 6 + self.E), dtype=torch.float32, device=self.dev)
        labs = torch.empty((B, self.num_point), dtype=torch.int64, device=self.dev)
        info = torch.empty((B, 4), dtype=torch.int32, device=self.dev)
        sel = torch.empty((B, self.num_point), dtype=torch.int32, device=self.dev) if want_indices else None
        with torch.cuda.device(self.dev):
            rc = lib.pn2_sample_blocks_multi(self.table.data_ptr(), len(self.samplers),
                                             ids.ctypes.data if on_host else rooms.data_ptr(), 1 if on_host else 0, self.E, self.block_size,
                                             self.num_point, 1024, int(seed) & (2 ** 64 - 1), B, feats.data_ptr(), labs.data_ptr(),
                                             info.data_ptr(), None if sel is None else sel.data_ptr(),
                                             torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "pn2_sample_blocks_multi")
        return (feats, labs, info, sel) if want_indices else (feats, labs, info)


class DeviceSceneTiler:
    """SceneTiler with the scene resident on the device (csrc/pn2_tiler.hip): the grid index is built
    once on the host and uploaded, the window table (a few hundred rows) is enumerated on the host with the reference's
    float64 arithmetic, and the points of every window are found, topped up, shuffled and laid out as blocks by two
    kernel launches and one small device -> host copy (the window populations, which size the output).
    tile(seed): randomness from keyed pseudo-random permutations of `seed` -- same seed, same tiling on every rank.
    tile_exact(rng): replays numpy's choice() / shuffle() call sequence of the reference from the window populations and
    yields the reference's blocks bit for bit (given the same global numpy RNG state).
    Both return device tensors (data [nb, bp, 6+E] float32, labels [nb, bp] int64, weights [nb, bp] float32, point
    indices [nb, bp] int64) that infer_scene() consumes without a host round trip."""

    def __init__(self, points, labels, extra=(), feature_name=(), labelweights=None, block_points=4096, block_size=1.0,
                 stride=0.5, padding=0.001, device="cuda"):
        import torch
        self.torch = torch
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :3])
        self.P = pts.shape[0]
        self.block_points, self.block_size, self.stride, self.padding = int(block_points), float(block_size), float(stride), float(padding)
        self.cmin, self.cmax = pts.min(axis=0), pts.max(axis=0)
        gi = GridIndex(pts[:, :2], cell=block_size / 4.0)
        self.grid = gi
        dev = torch.device(device)
        self.dev = dev
        self.xyz = torch.from_numpy(pts).to(dev)
        self.order = torch.from_numpy(gi.order.astype(np.int32)).to(dev)
        self.cell_start = torch.from_numpy(gi.start.astype(np.int32)).to(dev)
        lab = np.asarray(labels).astype(np.int64)
        self.labels = torch.from_numpy(lab).to(dev)
        lw = np.ones(int(lab.max()) + 1) if labelweights is None else np.asarray(labelweights, dtype=np.float64)
        self.num_classes = int(lw.shape[0])
        self.labelweights = torch.from_numpy(lw.astype(np.float32)).to(dev)
        cols = []
        for e, name in zip(extra, feature_name):
            e = np.asarray(e, dtype=np.float64)
            cols.append((e / 255 if name in ("red", "blue", "green") else e).astype(np.float32))      # :233-234
        self.E = len(cols)
        self.extra = torch.from_numpy(np.stack(cols)).to(dev) if cols else None
        self.windows, self.centres = window_table(self.cmin, self.cmax, self.block_size, self.stride, self.padding)
        self._members = None                         # (window table, counts, offsets, member lists): geometry only, built once

    def _lib(self):
        from . import _lib
        return _lib, _lib.load()

    def _find_members(self, ascending=False):
        """Window populations and member lists (independent of the seed).  ascending: every window's list in ascending point
        index -- what np.where yields in the reference; only the exact replay needs it."""
        torch = self.torch
        if self._members is not None and (self._members["ascending"] or not ascending):
            return self._members
        L, lib = self._lib()
        gi = self.grid
        st = torch.cuda.current_stream(self.dev).cuda_stream
        W = self.windows.shape[0]
        win = torch.from_numpy(self.windows).to(self.dev)
        counts = torch.empty(W, dtype=torch.int32, device=self.dev)
        with torch.cuda.device(self.dev):
            rc = lib.pn2_tile_windows(self.xyz.data_ptr(), self.order.data_ptr(), self.cell_start.data_ptr(), float(gi.origin[0]),
                                      float(gi.origin[1]), float(gi.cell), gi.nx, gi.ny, win.data_ptr(), W, None, counts.data_ptr(),
                                      None, st)
        L.check(rc, "pn2_tile_windows")
        cnt = counts.cpu().numpy().astype(np.int64)             # the one host sync of a tiling: it sizes the output
        keep = np.nonzero(cnt > 0)[0]                           # `if point_idxs.size == 0: continue` (:204)
        cnt = cnt[keep]
        win_k = torch.from_numpy(np.ascontiguousarray(self.windows[keep])).to(self.dev)
        off = np.concatenate(([0], np.cumsum(cnt)))
        nblk = (cnt + self.block_points - 1) // self.block_points
        boff = np.concatenate(([0], np.cumsum(nblk)))
        members = torch.empty(int(off[-1]), dtype=torch.int32, device=self.dev)
        moff = torch.from_numpy(off[:-1].copy()).to(self.dev)
        if keep.size:
            with torch.cuda.device(self.dev):
                rc = lib.pn2_tile_windows(self.xyz.data_ptr(), self.order.data_ptr(), self.cell_start.data_ptr(), float(gi.origin[0]),
                                          float(gi.origin[1]), float(gi.cell), gi.nx, gi.ny, win_k.data_ptr(), int(keep.size),
                                          moff.data_ptr(), None, members.data_ptr(), st)
            L.check(rc, "pn2_tile_windows")
        if ascending and keep.size:
            wid = torch.repeat_interleave(torch.arange(keep.size, device=self.dev), torch.from_numpy(cnt).to(self.dev))
            key = (wid.to(torch.int64) << 32) | members.to(torch.int64)
            members = (torch.sort(key)[0] & 0xFFFFFFFF).to(torch.int32)
        self._members = {"ascending": ascending, "keep": keep, "cnt": cnt, "off": off, "boff": boff, "members": members, "moff": moff,
                         "counts": torch.from_numpy(cnt.astype(np.int32)).to(self.dev),
                         "centre": torch.from_numpy(np.ascontiguousarray(self.centres[keep])).to(self.dev),
                         "boff_dev": torch.from_numpy(boff).to(self.dev)}
        return self._members

    def _fill(self, m, srcpos, seed):
        import ctypes
        torch = self.torch
        L, lib = self._lib()
        nb, bp, F = int(m["boff"][-1]), self.block_points, 6 + self.E
        data = torch.empty((nb, bp, F), dtype=torch.float32, device=self.dev)
        labels = torch.empty((nb, bp), dtype=torch.int64, device=self.dev)
        weight = torch.empty((nb, bp), dtype=torch.float32, device=self.dev)
        index = torch.empty((nb, bp), dtype=torch.int64, device=self.dev)
        cm = (ctypes.c_double * 3)(*[float(v) for v in self.cmax])
        with torch.cuda.device(self.dev):
            rc = lib.pn2_tile_fill(self.xyz.data_ptr(), None if self.extra is None else self.extra.data_ptr(), self.labels.data_ptr(),
                                   self.labelweights.data_ptr(), self.P, self.E, self.num_classes, cm, m["members"].data_ptr(),
                                   m["moff"].data_ptr(), m["counts"].data_ptr(), m["centre"].data_ptr(), m["boff_dev"].data_ptr(),
                                   int(m["keep"].size), nb, bp, None if srcpos is None else srcpos.data_ptr(),
                                   int(seed) & (2 ** 64 - 1), data.data_ptr(), labels.data_ptr(), weight.data_ptr(), index.data_ptr(),
                                   torch.cuda.current_stream(self.dev).cuda_stream)
        L.check(rc, "pn2_tile_fill")
        return data, labels, weight, index

    def tile(self, seed=0):
        return self._fill(self._find_members(), None, seed)

    def tile_exact(self, rng=np.random):
        """The reference's own random stream: per non-empty window rng.choice(population, top-up, replace) then
        rng.shuffle over the window's slots (:209-212) -- both consume the stream as a function of the sizes only, so the
        host draws POSITIONS and the device gathers."""
        m = self._find_members(ascending=True)
        bp = self.block_points
        parts = []
        for c in m["cnt"]:
            c = int(c)
            size = int(np.ceil(c / bp)) * bp
            fill = size - c
            rep = rng.choice(c, fill, replace=False if fill <= c else True)
            src = np.concatenate((np.arange(c), rep))
            rng.shuffle(src)
            parts.append(src)
        srcpos = self.torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(self.dev) if parts else None
        return self._fill(m, srcpos, 0)


class VotePool:
    """vote_label_pool of modelTesting (localfunctions.py:373-403) kept on the device as int32
    [num_points, num_classes]; add() scatters one vote per (point, arg-max class) with the HIP
    kernel behind pn2_add_vote."""

    def __init__(self, num_points, num_classes, device):
        import torch
        self.torch = torch
        self.pool = torch.zeros((num_points, num_classes), dtype=torch.int32, device=device)

    def add(self, logp=None, point_idx=None, weight=None, pred_label=None):
        from . import _lib
        from .ops import _dev, _err_word, _ptr, _stream
        torch = self.torch
        dev = _dev(logp, pred_label, point_idx, weight, self.pool)
        lib = _lib.load()
        P, C = self.pool.shape
        idx = point_idx.to(torch.int64).contiguous()
        M = idx.numel()
        lp = None if logp is None else logp.detach().to(torch.float32).contiguous()
        pl = None if pred_label is None else pred_label.to(torch.int64).contiguous()
        w = None if weight is None else weight.to(torch.float32).contiguous()
        if lp is not None and lp.numel() != M * C:
            raise ValueError("logp must hold %d x %d values" % (M, C))
        with torch.cuda.device(dev):
            rc = lib.pn2_add_vote(_ptr(lp), _ptr(pl), _ptr(idx), _ptr(w), M, C, P, _ptr(self.pool), _ptr(_err_word(dev)),
                                  _stream(dev))
        _lib.check(rc, "pn2_add_vote")

    def all_reduce(self, group=None):
        """Sum the pools of all ranks (one collective per scene): every rank voted on its own
        sub-batches of the scene, integer votes add exactly in any order.  No-op without a process group."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.pool, op=dist.ReduceOp.SUM, group=group)
        return self

    def labels(self):
        """np.argmax(vote_label_pool, 1) of localfunctions.py:405 (first maximum wins)."""
        return self.torch.argmax(self.pool, dim=1)


def shard_batches(num_blocks, batch_size, rank=0, world=1):
    """Start offsets of the sub-batches (localfunctions.py:386-390 walks range(s_batch_num)) this rank runs when a
    scene is sharded over `world` ranks: sub-batch i goes to rank i % world."""
    return list(range(0, num_blocks, batch_size))[rank::world]


# PN2_INFER_TWO_GRAPHS=0: the forked single graph of rounds 1-2 (the pyramid of the next sub-batch as a branch of the forward's graph)
_TWO_GRAPHS = os.environ.get("PN2_INFER_TWO_GRAPHS", "1") == "1"
# PN2_INFER_PLACE=0: the geometry graph packs the whole pyramid into a temporary and copies it into the static buffer
_PLACE = os.environ.get("PN2_INFER_PLACE", "1") == "1"
# PN2_INFER_LATE_SIDE_ENQUEUE=0: the next sub-batch's geometry graph is enqueued ahead of the forward behind a cross-stream wait
# instead of behind the forward and a host wait for the previous one (BlockInferencer.run)
_LATE_SIDE_ENQUEUE = os.environ.get("PN2_INFER_LATE_SIDE_ENQUEUE", "1") == "1"


class BlockInferencer:
    """Eval-mode forward passes over fixed-shape sub-batches [B, C, N] as ONE replayed hipGraph: while sub-batch i runs
    through the MLP stacks, the FPS / ball-query / 3-NN pyramid of sub-batch i+1 (with the first level's grouped rows and
    the laid-out input) is computed on a parallel branch of the same graph and handed over with one copy -- the scheme of
    train.SemSegTrainer, forward only.  BatchNorm is frozen (eval), so blocks are independent: a short last sub-batch is
    padded with copies of its last block.  The graph reads the weights through the pointers they had at capture and the
    eval-mode BatchNorm coefficients from the tensors cached on the modules: run() refreshes those in place from the
    current weights and running statistics first (mlp.refresh_eval_coefficients), so an engine kept across further
    training replays the trained model; if the parameters were re-homed since (train.FlatAdam moves them into one flat buffer when a trainer is created), run() notices and
    captures again."""

    def __init__(self, model, batch_size, channels, num_point):
        import torch
        self.torch = torch
        self.model = model.eval()
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise RuntimeError("BlockInferencer replays hipGraphs: the model must live on a HIP device")
        self.B = int(batch_size)
        self.cur_x = torch.zeros((self.B, channels, num_point), dtype=torch.float32, device=self.dev)
        self.next_x = torch.zeros_like(self.cur_x)
        self._side = torch.cuda.Stream(device=self.dev)
        self._graph = None
        self._captured_ptrs = None
        self.logp = None

    def __del__(self):
        # a captured graph must not be destroyed while a replay of it is still running (the geometry graph runs on the side
        # stream, behind the caller's back): wait for the device before the graphs go
        try:
            from . import ops
            if not ops.capturing():                               # (a device wait is illegal while a stream captures)
                self.torch.cuda.synchronize(self.dev)
        except Exception:
            pass

    def _param_ptrs(self):
        return [t.data_ptr() for t in list(self.model.parameters()) + list(self.model.buffers())]

    def _geometry_of(self, x):
        prepared = self.model.prepare_input(x, None)
        return self.model.compute_geometry(prepared=prepared, group_first=True, for_backward=False) + [prepared[0], prepared[1]]

    def _pack(self, geo):
        from .train import pack_segments
        return pack_segments(geo, self._pads)

    def _fill(self, flat, views):
        """The pyramid of `next_x` into the static buffer `flat` (whose tensor views are `views`): the first level's grouped rows
        -- nearly all of its bytes -- are written by the query launch itself (ops.place_next_grouped), the small tensors in front
        of and behind them are concatenated straight into the buffer (train.pack_segments(out=...)): no packed temporary of
        the whole pyramid and no copy of it (PN2_INFER_PLACE=0: both, as before)."""
        from . import ops
        from .train import pack_segments
        sizes = [0 if v is None else v.numel() * v.element_size() for v in views]
        big = max(range(len(views)), key=lambda i: sizes[i])
        view = views[big]
        placeable = _PLACE and view is not None and view.dtype == self.torch.float32 and view.dim() == 4
        if placeable:
            ops.place_next_grouped(view)
        try:
            geo = self._geometry_of(self.next_x)
        finally:
            ops.place_next_grouped(None)                          # an offer nobody took must not reach an unrelated call
        if placeable and geo[big] is not None and geo[big].data_ptr() == view.data_ptr():
            o0 = view.data_ptr() - flat.data_ptr()
            pad = 0 if self._pads[big] is None else self._pads[big].numel()
            o1 = o0 + sizes[big] + pad
            if big > 0:
                pack_segments(geo[:big], self._pads[:big], out=flat[:o0])
            if big + 1 < len(geo):
                pack_segments(geo[big + 1:], self._pads[big + 1:], out=flat[o1:])
            if pad:
                flat[o1 - pad:o1].zero_()
        else:
            flat.copy_(self._pack(geo))

    def _capture(self):
        from . import ops
        with ops.capture_region():                      # no garbage collection inside a capture (ops.capture_region)
            self._capture_graphs()

    def _capture_graphs(self):
        torch = self.torch
        with torch.no_grad():
            for _ in range(2):                                   # lazy initialisations, eval coefficients (cached)
                self.model(self.cur_x)
            first = self._geometry_of(self.cur_x)
            self._pads, off = [], 0
            for t in first:
                nbytes = 0 if t is None else t.numel() * t.element_size()
                pad = (-nbytes) % 16
                self._pads.append(torch.zeros(pad, dtype=torch.uint8, device=self.dev) if pad else None)
                off += nbytes + pad
            self._flat = torch.empty(off, dtype=torch.uint8, device=self.dev)
            self._flat.copy_(self._pack(first))
            self._cur, off = [], 0
            for t, padt in zip(first, self._pads):
                if t is None:
                    self._cur.append(None)
                    continue
                nbytes = t.numel() * t.element_size()
                self._cur.append(self._flat[off:off + nbytes].view(t.dtype).view(t.shape))
                off += nbytes + (0 if padt is None else padt.numel())
            torch.cuda.synchronize(self.dev)
            self._captured_ptrs = self._param_ptrs()
            self._two = None
            if _TWO_GRAPHS:
                # The pyramid of the next sub-batch as a hipGraph of its OWN on the side stream and two forward graphs that read
                # two pyramid buffers in turn: a forked graph keeps only one branch on its launch stream, and it was the
                # forward that moved to another queue (train.py, _SEPARATE_GEOMETRY_GRAPH); no hand-over copy either.
                flat2 = self._flat.clone()
                views2, off = [], 0
                for t, padt in zip(first, self._pads):
                    if t is None:
                        views2.append(None)
                        continue
                    nbytes = t.numel() * t.element_size()
                    views2.append(flat2[off:off + nbytes].view(t.dtype).view(t.shape))
                    off += nbytes + (0 if padt is None else padt.numel())
                bufs, views = [self._flat, flat2], [self._cur, views2]
                main = torch.cuda.current_stream()
                geos, fwds, logps = [], [], []
                gpool = None
                for p in (0, 1):
                    self._side.wait_stream(main)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._side, **({} if gpool is None else {"pool": gpool})):
                        self._fill(bufs[1 - p], views[1 - p])                              # fills the OTHER buffer
                    gpool = g.pool()
                    main.wait_stream(self._side)
                    torch.cuda.synchronize(self.dev)
                    geos.append(g)
                fpool = None
                for p in (0, 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, **({} if fpool is None else {"pool": fpool})):
                        logp, _ = self.model(self.cur_x, geometry=views[p][:-2], prepared=(views[p][-2], views[p][-1]))
                    fpool = g.pool()
                    fwds.append(g)
                    logps.append(logp)
                self._two = {"geo": geos, "fwd": fwds, "logp": logps, "bufs": bufs, "p": 0, "ready": torch.cuda.Event(),
                             "inputs": torch.cuda.Event(), "ends": [torch.cuda.Event(), torch.cuda.Event()]}
                self._graph = fwds[0]
                self.logp = logps[0]
                return
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                main = torch.cuda.current_stream()
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):              # fork: the next sub-batch's pyramid
                    new_flat = self._pack(self._geometry_of(self.next_x))
                geo = self._cur
                logp, _ = self.model(self.cur_x, geometry=geo[:-2], prepared=(geo[-2], geo[-1]))
                self.logp = logp
                self._side.wait_stream(main)                     # the forward no longer reads the current pyramid
                with torch.cuda.stream(self._side):
                    self._flat.copy_(new_flat)
                main.wait_stream(self._side)                     # join

    def run(self, batches, consume):
        """batches: a sequence of [b, C, N] arrays / tensors, b <= batch_size (channel-first, like the reference's
        `torch_data.transpose(2, 1)`, localfunctions.py:398); consume(i, logp[:b]) is called for each with the
        log-probabilities [b, N, classes] -- a view of a static buffer, valid until the next replay is enqueued (enqueue
        what reads it on the current stream, as VotePool.add does)."""
        torch = self.torch
        batches = list(batches)
        if not batches:
            return
        with torch.no_grad():
            def load(dst, src):
                src = torch.as_tensor(src, dtype=torch.float32, device=self.dev)
                b = src.shape[0]
                dst[:b].copy_(src)
                if b < self.B:
                    dst[b:].copy_(src[b - 1:b].expand(self.B - b, -1, -1))
                return b
            if self._graph is not None and self._captured_ptrs != self._param_ptrs():
                self._graph = None                                # the parameters live elsewhere now: the graph reads stale memory
            if self._graph is None:
                self._capture()
            # the captured forwards read the eval-mode BatchNorm scale / shift from the tensors cached on the modules; if the
            # model was trained since they were computed they are recomputed here, in place (a key compare per BatchNorm when
            # nothing changed) -- an engine kept across training would otherwise mix new weights with stale coefficients
            from . import mlp
            mlp.refresh_eval_coefficients(self.model)
            two = self._two
            main = torch.cuda.current_stream()
            if two is not None:
                main.wait_event(two["ready"])                     # (a geometry graph of an earlier run)
            load(self.next_x, batches[0])
            (self._flat if two is None else two["bufs"][two["p"]]).copy_(self._pack(self._geometry_of(self.next_x)))   # the first pyramid
            prev_end = None
            for i, blocks in enumerate(batches):
                nxt = batches[i + 1] if i + 1 < len(batches) else blocks
                b = blocks.shape[0]
                if two is None:
                    load(self.next_x, nxt)
                    self._graph.replay()
                    consume(i, self.logp[:b])
                    continue
                p = two["p"]
                if i > 0:
                    main.wait_event(two["ready"])                 # the side graph has read next_x and filled buffer p
                if not _LATE_SIDE_ENQUEUE:
                    load(self.next_x, nxt)
                    self._side.wait_stream(main)
                    two["fwd"][p].replay()                        # reads pyramid buffer p
                    with torch.cuda.stream(self._side):
                        two["geo"][p].replay()                    # the pyramid of next_x into buffer 1 - p
                        two["ready"].record()
                else:
                    # the side stream's work is enqueued BEHIND this forward and after a host wait for the previous one: its
                    # dependency (the previous forward read the buffer it fills) is then satisfied when it is enqueued, and the
                    # side queue never sits blocked at a cross-stream barrier -- a graph runs measurably longer beside a queue
                    # that does (train.SemSegTrainer._enqueue_geometry)
                    two["inputs"].record(main)                    # whatever produced `nxt` is in the main queue up to here
                    two["fwd"][p].replay()                        # reads pyramid buffer p
                    end = two["ends"][i & 1]
                    end.record(main)
                    if prev_end is not None:
                        prev_end.synchronize()
                    prev_end = end
                    if i + 1 < len(batches):                      # (no pyramid behind the last sub-batch: nothing stays in flight)
                        with torch.cuda.stream(self._side):
                            self._side.wait_event(two["inputs"])
                            load(self.next_x, nxt)
                            two["geo"][p].replay()                # the pyramid of next_x into buffer 1 - p
                            two["ready"].record()
                self.logp = two["logp"][p]
                consume(i, self.logp[:b])
                two["p"] = p ^ 1


def scene_metrics(pred_label, labels, num_classes):
    """Per-class seen / correct / union counters of one scene and what modelTesting logs from them
    (localfunctions.py:409-421, 463-479): IoU = correct / (union + 1e-6) per class, `mIoU` their mean over ALL classes
    (:477), `scene_mIoU` the mean over the classes that occur in the scene (:418-419), accuracies (:478-479).
    pred_label / labels: tensors or arrays [P]."""
    import torch
    pred = torch.as_tensor(pred_label).reshape(-1).to(torch.int64)
    lab = torch.as_tensor(labels).reshape(-1).to(torch.int64).to(pred.device)
    C = int(num_classes)
    seen = torch.bincount(lab, minlength=C)[:C]
    hit = pred == lab
    correct = torch.bincount(lab[hit], minlength=C)[:C]
    union = seen + torch.bincount(pred, minlength=C)[:C] - correct           # |pred == l or label == l|
    seen, correct, union = (t.cpu().numpy().astype(np.float64) for t in (seen, correct, union))
    iou = correct / (union + 1e-6)
    return {"class_seen": seen, "class_correct": correct, "class_union": union, "IoU": iou, "mIoU": float(iou.mean()),
            "scene_mIoU": float(iou[seen != 0].mean()) if (seen != 0).any() else 0.0,
            "avg_class_acc": float((correct / (seen + 1e-6)).mean()), "accuracy": float(correct.sum() / (seen.sum() + 1e-6))}


def infer_scene(model, data_room, index_room, sample_weight, num_points, num_classes, batch_size=32, num_votes=1,
                retile=None, group=None, graphs=False, return_votes=False, engine=None):
    """Whole-scene voting inference (localfunctions.py:375-405): run the network over the scene's
    blocks in sub-batches, vote on the device, return the per-point predicted label tensor.
    `retile`, if given, is called before every vote round after the first to re-draw the blocks
    (the reference re-tiles per vote, :377; with several ranks it must return the same tiling on every rank).
    Under torch.distributed (one process per GPU) the sub-batches of the scene are sharded over the ranks and the
    int32 vote pools are summed with ONE all-reduce per scene (RCCL over xGMI when the backend is "nccl"); every
    rank returns the full label tensor.
    graphs=True (HIP device): the sub-batches run through a BlockInferencer (one replayed graph, the next sub-batch's
    pyramid prefetched); engine = a BlockInferencer of this model and sub-batch shape to reuse (capturing the graph costs
    about as much as a few hundred blocks: keep one per model across scenes)."""
    import torch
    import torch.distributed as dist
    dev = next(model.parameters()).device
    votes = VotePool(num_points, num_classes, dev)
    model.eval()
    if engine is None and graphs and dev.type == "cuda" and hasattr(model, "compute_geometry"):
        engine = BlockInferencer(model, batch_size, data_room.shape[2], data_room.shape[1])
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    with torch.no_grad():
        for v in range(num_votes):
            if v > 0 and retile is not None:
                data_room, _, sample_weight, index_room = retile()
            starts = shard_batches(data_room.shape[0], batch_size, rank, world)
            if engine is not None:
                def vote(i, logp, starts=starts, index_room=index_room, sample_weight=sample_weight):
                    s0 = starts[i]
                    votes.add(logp=logp, point_idx=torch.as_tensor(index_room[s0:s0 + batch_size], device=dev),
                              weight=torch.as_tensor(sample_weight[s0:s0 + batch_size], dtype=torch.float32, device=dev))
                engine.run([torch.as_tensor(data_room[s0:s0 + batch_size], dtype=torch.float32, device=dev).transpose(2, 1)
                            for s0 in starts], vote)
                continue
            for s in starts:
                x = torch.as_tensor(data_room[s:s + batch_size], dtype=torch.float32, device=dev).transpose(2, 1)
                logp, _ = model(x)
                votes.add(logp=logp, point_idx=torch.as_tensor(index_room[s:s + batch_size], device=dev),
                          weight=torch.as_tensor(sample_weight[s:s + batch_size], dtype=torch.float32, device=dev))
    votes.all_reduce(group)
    return (votes.labels(), votes.pool) if return_votes else votes.labels()
