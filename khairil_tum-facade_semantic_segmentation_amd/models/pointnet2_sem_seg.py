"""pointnet2_sem_seg on the HIP operator surface.

Mirror of the reference's models/pointnet2_sem_seg.py:6-50 (the caller of the hot path, SURVEY.md
8a-8): same constructor, same submodule names (=> identical state_dict keys), same outputs
(log-probabilities [B,N,classes], l4 features [B,512,16]).  Activations stay channel-last
between levels so no tensor is re-laid-out on the way through the network."""
import os

import torch.nn as nn
import torch.nn.functional as F

from .. import head, ops
from . import pointnet2_utils as _utils
from .pointnet2_utils import PointNetFeaturePropagation, PointNetSetAbstraction

_INVERT_GROUPING = os.environ.get("PN2_INVERT_GROUPING", "0") == "1"
# PN2_GEOMETRY_MERGED_LAUNCHES=1: the four interpolation levels' 3-NN tables and their transposed tables as ONE launch each
# (2: the transposed tables only, 3: the 3-NN tables only).  Measured on one box, 60 steps, twice each: 2.534 ms per level
# (default), transposed tables merged 2.533-2.539, 3-NN merged 2.545, both 2.551 -- a shorter geometry chain buys nothing
# (it has slack) and one bigger launch beside the main branch costs it more than four small ones.
_MERGED_LAUNCHES = os.environ.get("PN2_GEOMETRY_MERGED_LAUNCHES", "0")
# the head's dropout inside the conv2 kernels (PN2_FUSED_DROPOUT=0: torch's F.dropout in front of them, for A/B runs)
_FUSED_DROPOUT = os.environ.get("PN2_FUSED_DROPOUT", "1") == "1"
# gradients of a level's features (grouping of the next level + skip connection) summed inside the grouping backward
_SKIP_IN_SCATTER = os.environ.get("PN2_SKIP_IN_SCATTER", "1") == "1"
# conv1 / bn1 / relu of the head as the last layers of fp1's stack (fp1's activated output is never written, one
# top-of-stack BatchNorm-backward reduction less); PN2_HEAD_IN_FP1=0: a stack of its own, for A/B runs
_HEAD_IN_FP1 = os.environ.get("PN2_HEAD_IN_FP1", "1") == "1"

# (npoint, radius, nsample, mlp) per set-abstraction level; reference :9-12
SA_LEVELS = ((1024, 0.1, 32, (32, 32, 64)), (256, 0.2, 32, (64, 64, 128)),
             (64, 0.4, 32, (128, 128, 256)), (16, 0.8, 32, (256, 256, 512)))
# (name, in_channel, mlp) per feature-propagation level; reference :13-16
FP_LEVELS = (("fp4", 768, (256, 256)), ("fp3", 384, (256, 256)), ("fp2", 320, (256, 128)),
             ("fp1", 128, (128, 128, 128)))


class get_model(nn.Module):
    def __init__(self, num_classes, num_extra_features):
        super().__init__()
        cin = 6 + 3 + num_extra_features                 # l0 features are the whole input incl. xyz (:23-26)
        for i, (npoint, radius, nsample, mlp) in enumerate(SA_LEVELS, start=1):
            setattr(self, "sa%d" % i, PointNetSetAbstraction(npoint, radius, nsample, cin, list(mlp), False))
            cin = mlp[-1] + 3
        for name, cin, mlp in FP_LEVELS:
            setattr(self, name, PointNetFeaturePropagation(cin, list(mlp)))
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def prepare_input(self, xyz, angles=None):
        """The step's input preparation as one kernel (ops.input_blocks): xyz [B, C, N] channel-first -> (rows [B,N,C],
        coordinates [B,N,3]), with the per-block rotation about the up axis of the reference loop (localfunctions.py:206)
        when `angles` [B] is given.  Pass the result as `prepared=` to compute_geometry() / forward()."""
        return ops.input_blocks(xyz, True, angles)

    def compute_geometry(self, xyz=None, prepared=None, group_first=False, for_backward=True):
        """Everything in the forward pass that depends on the input coordinates only: the FPS /
        ball-query pyramid of the four SA levels and the 3-NN tables of the four FP levels
        (SURVEY.md 3.3).  Returns a flat list of tensors that forward(geometry=...) consumes; it
        can be computed ahead of time (another stream, one batch early) because no learned
        quantity feeds it.
        group_first (with `prepared`): the first level's grouped rows [B,S,K,12] too -- its features are the network input,
        so the launch that finds the indices gathers the rows as well (the fused query + group kernel of DESIGN 4.2)
        and forward(geometry=...) starts at the first GEMM; slot [32] of the list (None when the shape is outside the
        planned path).
        for_backward=False (inference): the transposed tables that only the atomic-free backward reads are left out."""
        cur = prepared[1] if prepared is not None else xyz.permute(0, 2, 1)[:, :, :3].contiguous()
        levels = [cur]
        out, inv = [], []
        grouped1 = None
        for i, sa in enumerate((self.sa1, self.sa2, self.sa3, self.sa4)):
            if i == 0 and group_first and prepared is not None:
                new_xyz, idx, grouped1 = sa.geometry(levels[-1], points=prepared[0])
            else:
                new_xyz, idx = sa.geometry(levels[-1])
            out += [new_xyz, idx]
            # A transposed index for the grouping backward is possible too (ops.group_points(inv=...): atomic-free, a fixed
            # summation order).  With a key row's list split over four lane slices the gather-sums themselves beat the
            # float atomics (levels 2-4: 22 + 18 + 14 us against 37 + 20 + 14), but the three extra pn2_invert_index
            # launches (one 1024-thread workgroup per block, 50-100 us each) put 0.26 ms more on the geometry branch and
            # its kernels slow the main branch's: +58 us per step (PN2_INVERT_GROUPING=1, profiles/r03).  Interpolation only.
            pair = ops.invert_index(idx, levels[-1].shape[1]) if (i > 0 and _INVERT_GROUPING and for_backward) else None
            inv += list(pair) if pair is not None else [None, None]
            levels.append(new_xyz)
        idx3s = []
        # the four levels' nearest-neighbour tables in one launch (they need the levels' coordinates and nothing else)
        nn = ops.three_nn_many([(levels[lvl], levels[lvl + 1]) for lvl in (3, 2, 1, 0)]) if (levels[0].is_cuda and _MERGED_LAUNCHES in ("1", "3")) else \
            [ops.three_nn(levels[lvl], levels[lvl + 1]) for lvl in (3, 2, 1, 0)]
        for idx3, w3 in nn:
            out += [idx3, w3]
            idx3s.append(idx3)
        if for_backward:
            # the four transposed tables in one launch (one workgroup per block each: in a row they only add up latencies)
            keys = [levels[lvl + 1].shape[1] for lvl in (3, 2, 1, 0)]
            pairs = ops.invert_index_many(idx3s, keys) if (idx3s[0].is_cuda and _MERGED_LAUNCHES in ("1", "2")) else None
            if pairs is None:
                pairs = [ops.invert_index(i3, k) for i3, k in zip(idx3s, keys)]
            for pair in pairs:
                inv += list(pair) if pair is not None else [None, None]
        else:
            inv += [None, None] * 4
        # [0:8] SA, [8:16] FP, [16:24] SA inverses, [24:32] FP inverses, [32] grouped rows of level 1
        return out + inv + ([grouped1] if group_first else [])

    @staticmethod
    def _inverse(geometry, slot):
        if len(geometry) <= 16 or geometry[16 + 2 * slot] is None:
            return ()
        return ((geometry[16 + 2 * slot], geometry[17 + 2 * slot]),)

    def forward(self, xyz, geometry=None, prepared=None):
        """xyz [B, 3+3+extra, N] -> (log_softmax [B,N,classes], l4_points [B,512,16]).  prepared = prepare_input(xyz,
        angles): the (rotated) rows and coordinates to run on instead of xyz itself."""
        if prepared is not None:
            pts, xyz3 = prepared
        else:
            pts, xyz3 = ops.input_blocks(xyz, True, None)   # [B,N,C] rows and [B,N,3] coordinates in one pass
        geo = [xyz3]
        feat = [pts]
        for i, sa in enumerate((self.sa1, self.sa2, self.sa3, self.sa4)):
            pre = None if geometry is None else (geometry[2 * i], geometry[2 * i + 1]) + self._inverse(geometry, i)
            if i == 0 and pre is not None and len(geometry) > 32 and geometry[32] is not None and not pts.requires_grad:
                g, f = sa.forward_cl(geo[-1], feat[-1], geometry=pre, grouped=geometry[32])
            elif i > 0 and pre is not None and _SKIP_IN_SCATTER:
                # feat[i] feeds this level's grouping AND a feature-propagation skip: route the skip through the grouping
                # op's second output, so that both gradients meet inside its scatter (no zero fill, no add kernel)
                # The skip's only consumer is a feature-propagation stack whose backward (mlp._MLPStack) returns a tensor
                # it has just allocated, so the scatter may add onto that tensor itself (no copy: 3 x 11 us per step);
                # with the torch-op A/B path nothing is known about who else holds the gradient.
                g, f, feat[-1] = sa.forward_cl(geo[-1], feat[-1], geometry=pre,
                                               skip=True if _utils._TORCH_MLP else "inplace")
            else:
                g, f = sa.forward_cl(geo[-1], feat[-1], geometry=pre)
            geo.append(g)
            feat.append(f)
        up = feat[4]
        for j, (lvl, fp) in enumerate(zip((3, 2, 1, 0), (self.fp4, self.fp3, self.fp2, self.fp1))):
            pre = None if geometry is None else (geometry[8 + 2 * j], geometry[9 + 2 * j]) + self._inverse(geometry, 4 + j)
            # the head's conv1 -> bn1 -> relu (:36) reads the last level's output and nothing else does: one stack
            tail = ([self.conv1], [self.bn1]) if (lvl == 0 and _HEAD_IN_FP1 and not _utils._TORCH_MLP) else None
            up = fp.forward_cl(geo[lvl], geo[lvl + 1], feat[lvl] if lvl else None, up, nn=pre, tail=tail)   # :31-34
        if _utils._TORCH_MLP:                            # A/B switch: the head through torch ops
            h = up.permute(0, 2, 1)
            h = self.drop1(F.relu(self.bn1(self.conv1(h))))  # :36
            h = F.log_softmax(self.conv2(h), dim=1)          # :37-38
            return h.permute(0, 2, 1), feat[4].permute(0, 2, 1)
        B, N, C = up.shape
        if _HEAD_IN_FP1:
            h = up.reshape(B * N, C)                     # conv1 / bn1 / relu ran as the last layers of fp1's stack
        else:
            h = _utils._mlp(up.reshape(B * N, C), None, [self.conv1], [self.bn1])   # conv1 -> bn1 -> relu (:36)
        # dropout (:36) inside the head kernels: the keep-mask is regenerated from a device seed, never stored
        p = float(self.drop1.p) if self.training else 0.0
        if p > 0.0 and _FUSED_DROPOUT:
            logp = head.head_logits(h, self.conv2.weight, self.conv2.bias, drop_p=p)
        else:
            logp = head.head_logits(self.drop1(h), self.conv2.weight, self.conv2.bias)     # conv2 -> log_softmax (:37-38)
        return logp.view(B, N, -1), feat[4].permute(0, 2, 1)


class get_loss(nn.Module):                               # reference :44-50
    def forward(self, pred, target, trans_feat, weight):
        if _utils._TORCH_MLP:
            return F.nll_loss(pred, target, weight=weight)
        return head.nll_loss(pred, target, weight)
