"""Drop-in operator surface for the reference's models/pointnet2_utils.py, backed by the
gfx950 HIP library (include/pn2_hip.h).

Same names, argument order, tensor layouts ([B,N,C] for free functions, [B,C,N] for modules),
dtypes (int64 indices) and state_dict keys as the reference file, so models/pointnet2_sem_seg.py
and checkpoints written by either implementation are interchangeable.  What differs is how the
work is done: no [B,S,N] distance matrix, no sort, no python FPS loop -- see csrc/.

Reference lines cited per symbol; `start=` / fps_starts() expose the FPS start indices the
reference draws internally with torch.randint (:75)."""
import contextlib
from time import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import graphed, mlp, ops

_FPS_START_QUEUE = []


@contextlib.contextmanager
def fps_starts(starts):
    """Inject FPS start indices: each farthest_point_sample call inside the block pops one
    [B] tensor from `starts` instead of drawing torch.randint (parity tests, reproducibility)."""
    _FPS_START_QUEUE.extend(list(starts))
    try:
        yield
    finally:
        del _FPS_START_QUEUE[:]


def _next_start(device):
    if not _FPS_START_QUEUE:
        return None
    return torch.as_tensor(_FPS_START_QUEUE.pop(0), dtype=torch.long).to(device)


def timeit(tag, t):                                     # reference :7-9
    print("{}: {}s".format(tag, time() - t))
    return time()


def pc_normalize(pc):                                   # reference :11-17
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def square_distance(src, dst):
    """[B,N,3] x [B,M,3] -> [B,N,M], bit-identical to the reference's CPU result (:19-40)."""
    return ops.square_distance(src, dst)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] | [B,S,K] -> [B,S,(K,)C]  (:43-60); differentiable in points."""
    return ops.index_points(points, idx)


def farthest_point_sample(xyz, npoint, start=None):
    """xyz [B,N,3] -> centroids [B,npoint] int64  (:63-84)."""
    if start is None:
        start = _next_start(xyz.device)
    return ops.farthest_point_sample(xyz, npoint, start)


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> group_idx [B,S,nsample] int64: the nsample lowest indices within radius, tail padded
    with the first (:87-107)."""
    return ops.query_ball_point(radius, nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, start=None, pad_to=1):
    """-> new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D]  (:110-138).  pad_to > 1 (internal)
    rounds the last dimension up with zero columns."""
    if start is None:
        start = _next_start(xyz.device)
    # FPS, then ONE launch from (xyz, new_xyz, points) to (idx, grouped rows): the self-contained operator (a query
    # plan would be used once here and costs more to build than it saves)
    fps_idx, new_xyz = ops.farthest_point_sample_with_xyz(xyz, npoint, start)
    idx, new_points = ops.ball_query_group(radius, nsample, xyz, new_xyz, points, pad_to)
    if returnfps:
        grouped_xyz = ops.index_points(xyz, idx)
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """-> new_xyz [B,1,3] (zeros), new_points [B,1,N,3+D]  (:141-158)."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device, dtype=xyz.dtype)
    grouped = xyz.reshape(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.reshape(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


# PN2_TORCH_MLP=1 routes the conv/BN/ReLU stacks through torch ops (rocBLAS + ATen BatchNorm)
# instead of the MFMA kernels of csrc/pn2_mlp.hip: an A/B switch for benchmarks and tests.
_TORCH_MLP = os.environ.get("PN2_TORCH_MLP", "0") == "1"


def _mlp(x1, x2, convs, bns, pool_k=0):
    """[rows, K1] | [rows, K2] -> [rows(/pool_k), Cout] through the fused HIP stack.  x1 may carry
    zero pad columns beyond the first conv's input width (see ops.ball_query_group pad_to)."""
    if _TORCH_MLP:
        x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
        y = _pointwise_mlp(x[:, :convs[0].weight.shape[1]], convs, bns)
        return y.reshape(-1, pool_k, y.shape[-1]).max(dim=1)[0] if pool_k else y
    return mlp.mlp_stack(x1, x2, convs, bns, pool_k)


def _pointwise_mlp(x, convs, bns):
    """torch-op form of the stack (A/B reference): (1x1 conv -> BatchNorm -> ReLU) per layer on
    channel-last rows; BN statistics run over all rows, which is exactly BatchNorm2d over (B,K,S)
    / BatchNorm1d over (B,N)."""
    for conv, bn in zip(convs, bns):
        w = conv.weight.reshape(conv.weight.shape[0], -1)
        x = torch.addmm(conv.bias, x, w.t()) if conv.bias is not None else x @ w.t()
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        x = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                         bn.training or not bn.track_running_stats,
                         0.0 if bn.momentum is None else bn.momentum, bn.eps)
        x = F.relu(x)
    return x


def _before_replay(module):
    """Host-side state a captured stack reads through device memory, refreshed as the eager call would: the BatchNorm
    momentum words (the reference loop resets `momentum` every epoch, localfunctions.py:191-195) and, in eval mode, the
    cached scale / shift of the running statistics; a training pass rewrites those statistics behind torch's back."""
    for bn in module.mlp_bns:
        if bn.weight is not None:
            mlp.momentum_word(bn, bn.weight.device)
    if module.training:
        mlp.invalidate_eval_coefficients()
    else:
        mlp.refresh_eval_coefficients(module)


class PointNetSetAbstraction(nn.Module):                # reference :161-202
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        widths = [in_channel] + list(mlp)
        for ci, co in zip(widths[:-1], widths[1:]):
            self.mlp_convs.append(nn.Conv2d(ci, co, 1))
            self.mlp_bns.append(nn.BatchNorm2d(co))

    def geometry(self, xyz, start=None, points=None):
        """The part of the level that depends on xyz only: FPS centroids and
        ball-query indices.  xyz [B,N,3] -> (new_xyz [B,S,3], idx [B,S,K]).
        points [B,N,D] (features that need no gradient -- the network input): a third result, the grouped rows
        [B,S,K,pad4(3+D)] from the SAME launch that finds the indices (the planned query gathers them from the plan's
        packed rows); None when the shape is outside the planned path."""
        if "deep" in ops._LAB_SKIP and points is None and not getattr(self, "_lab_inside", False):
            self._lab_inside = True                      # lab switch (ops.lab_cached): levels without input rows, computed once
            try:
                return ops.lab_cached("deep", (tuple(xyz.shape), self.npoint), lambda: self.geometry(xyz, start))
            finally:
                self._lab_inside = False
        if start is None:
            start = _next_start(xyz.device)
        B, N, _ = xyz.shape
        if xyz.is_cuda and ops.plan_supported(B, N, self.npoint) and self.nsample <= 64:
            if points is not None:
                _, new_xyz, plan = ops.farthest_point_sample_plan(xyz, self.npoint, self.radius, points.shape[2], start)
                idx, grouped = ops.ball_query_group(self.radius, self.nsample, xyz, new_xyz, points, pad_to=4, plan=plan)
                return new_xyz, idx, grouped
            _, new_xyz, plan = ops.farthest_point_sample_plan(xyz, self.npoint, self.radius, 0, start)
            idx = ops.query_ball_point(self.radius, self.nsample, xyz, new_xyz, plan=plan)
            return (new_xyz, idx) if points is None else (new_xyz, idx, None)
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, self.npoint, start)
        idx = ops.query_ball_point(self.radius, self.nsample, xyz, new_xyz)
        return (new_xyz, idx) if points is None else (new_xyz, idx, None)

    def forward_cl(self, xyz, points, start=None, geometry=None, skip=False, grouped=None):
        """Channel-last form: xyz [B,N,3], points [B,N,D] -> new_xyz [B,S,3], feats [B,S,C'].
        `geometry` = a precomputed (new_xyz, idx[, inverse index]) tuple from geometry().
        skip (with `geometry`): a third result, `points` again for the caller's skip connection (ops.group_points
        with_skip: the two gradients of the points are then summed inside the grouping backward; skip="inplace": onto
        the skip gradient itself, which the caller states has no other reader).
        grouped (with `geometry`): the level's grouped rows, already built by geometry(points=...) -- no grouping pass
        here (points that need a gradient must not take this route: the rows are a constant to autograd)."""
        alias = points
        if self.group_all:
            new_xyz, grouped = sample_and_group_all(xyz, points)
        elif geometry is not None and grouped is not None:
            new_xyz = geometry[0]
        elif geometry is not None:
            new_xyz, idx = geometry[0], geometry[1]
            inv = geometry[2] if len(geometry) > 2 else None       # ops.invert_index(idx, N): atomic-free backward
            if skip and points is not None:
                grouped, alias = ops.group_points(xyz, new_xyz, points, idx, pad_to=4, inv=inv, with_skip=skip)
            else:
                grouped = ops.group_points(xyz, new_xyz, points, idx, pad_to=4, inv=inv)
        else:
            new_xyz, grouped = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points, start=start,
                                                pad_to=4)
        B, S, K, C = grouped.shape
        y = _mlp(grouped.reshape(B * S * K, C), None, self.mlp_convs, self.mlp_bns, pool_k=K)   # :196-200
        if skip:
            return new_xyz, y.reshape(B, S, -1), alias
        return new_xyz, y.reshape(B, S, -1)

    def _forward_rows(self, xyz, points, start):
        """The module's work on the caller's channel-first tensors, channel-last results: what a graph replays."""
        return self.forward_cl(xyz.permute(0, 2, 1), None if points is None else points.permute(0, 2, 1), start=start)

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,D',S]."""
        if not self.group_all and not _TORCH_MLP and graphed.usable(xyz, points):
            # repeated calls of one signature replay a captured forward / backward (graphed.py).  The FPS start indices are
            # an input of the graph, drawn here exactly as the eager path draws them (reference :75)
            start = _next_start(xyz.device)
            if start is None:
                start = torch.randint(0, xyz.shape[2], (xyz.shape[0],), dtype=torch.long, device=xyz.device)
            new_xyz, feats = graphed.call(self, self._forward_rows, (xyz, points, start), lambda: _before_replay(self))
            return new_xyz.permute(0, 2, 1), feats.permute(0, 2, 1)
        new_xyz, feats = self.forward_cl(xyz.permute(0, 2, 1), None if points is None else points.permute(0, 2, 1))
        return new_xyz.permute(0, 2, 1), feats.permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):             # reference :205-262
    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            widths = [in_channel + 3] + list(mlp)
            for ci, co in zip(widths[:-1], widths[1:]):
                convs.append(nn.Conv2d(ci, co, 1))
                bns.append(nn.BatchNorm2d(co))
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        xyz = xyz.permute(0, 2, 1)
        points = None if points is None else points.permute(0, 2, 1)
        _, new_xyz = ops.farthest_point_sample_with_xyz(xyz, self.npoint, _next_start(xyz.device))
        scales = []
        for radius, K, convs, bns in zip(self.radius_list, self.nsample_list, self.conv_blocks, self.bn_blocks):
            _, g = ops.ball_query_group(radius, K, xyz, new_xyz, points)
            if points is not None:                       # multi-scale variant orders [feats, xyz] (:248)
                g = torch.cat([g[..., 3:], g[..., :3]], dim=-1)
            B, S, _, C = g.shape
            y = _mlp(g.reshape(B * S * K, C), None, convs, bns, pool_k=K)
            scales.append(y.reshape(B, S, -1))
        return new_xyz.permute(0, 2, 1), torch.cat(scales, dim=-1).permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):            # reference :265-315
    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        widths = [in_channel] + list(mlp)
        for ci, co in zip(widths[:-1], widths[1:]):
            self.mlp_convs.append(nn.Conv1d(ci, co, 1))
            self.mlp_bns.append(nn.BatchNorm1d(co))

    def forward_cl(self, xyz1, xyz2, points1, points2, nn=None, tail=None):
        """Channel-last: xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,D1]|None, points2 [B,S,D2] -> [B,N,C'].
        `nn` = a precomputed (idx3, weight3[, inverse index]) tuple from ops.three_nn(xyz1, xyz2).
        tail = (convs, bns): further pointwise conv / BatchNorm / ReLU layers of the CALLER that consume this level's
        output and nothing else does (the segmentation head's conv1 / bn1 after the last level): run as more layers of
        the same stack -- this level's activated output is then never materialised, and its backward needs no
        top-of-stack reduction of its own."""
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:                                      # :293-294
            interpolated = points2.expand(B, N, points2.shape[-1])
        else:
            idx3, w3 = (nn[0], nn[1]) if nn is not None else ops.three_nn(xyz1, xyz2)   # :296-302
            inv = nn[2] if nn is not None and len(nn) > 2 else None    # ops.invert_index(idx3, S)
            interpolated = ops.three_interpolate(points2, idx3, w3, inv=inv)   # :303
        D2 = interpolated.shape[-1]
        convs, bns = list(self.mlp_convs), list(self.mlp_bns)
        if tail is not None:
            convs, bns = convs + list(tail[0]), bns + list(tail[1])
        if points1 is None:                             # :305-309 (the concat is never materialised)
            y = _mlp(interpolated.reshape(B * N, D2), None, convs, bns)
        else:
            y = _mlp(points1.reshape(B * N, -1), interpolated.reshape(B * N, D2), convs, bns)
        return y.reshape(B, N, -1)

    def _forward_rows(self, xyz1, xyz2, points1, points2):
        return (self.forward_cl(xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1),
                                None if points1 is None else points1.permute(0, 2, 1), points2.permute(0, 2, 1)),)

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N]|None, points2 [B,D2,S] -> [B,D',N]."""
        if not _TORCH_MLP and graphed.usable(xyz1, xyz2, points1, points2):
            (y,) = graphed.call(self, self._forward_rows, (xyz1, xyz2, points1, points2), lambda: _before_replay(self))
        else:
            (y,) = self._forward_rows(xyz1, xyz2, points1, points2)
        return y.permute(0, 2, 1)
