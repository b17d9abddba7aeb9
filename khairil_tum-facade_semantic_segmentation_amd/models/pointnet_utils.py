"""Plain PointNet encoder -- the pointwise-MLP-only control of BASELINE configs[4] (no set abstraction, no
sampling / grouping): same classes, constructor arguments, forward signatures and state_dict keys as the reference's
models/pointnet_utils.py (STN3d :10-43, STNkd :46-82, PointNetEncoder :85-133, feature_transform_reguliarzer
:136-142), so checkpoints are interchangeable.

The [conv 1x1 / linear -> BatchNorm -> ReLU] chains and the max over the points run through the same MFMA stack as
the set-abstraction MLPs (mlp.mlp_stack on channel-last rows, two-stage max-pool over the N points); what the stack
has no form for stays on torch / rocBLAS: the tiny per-block 3x3 and 64x64 transforms (torch.bmm), and the one
layer the reference leaves WITHOUT a ReLU before the max (bn3(conv3(x)), :123)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pointnet2_utils import _mlp


def _rows(x):
    """[B, C, N] channel-first -> contiguous rows [B*N, C]"""
    B, C, N = x.shape
    return x.permute(0, 2, 1).reshape(B * N, C)


class STN3d(nn.Module):                                   # reference :10-43
    def __init__(self, channel):
        super().__init__()
        self.conv1 = torch.nn.Conv1d(channel, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 128, 1)
        self.conv3 = torch.nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.k = 3

    def forward_rows(self, rows, B, N):
        """rows [B*N, C] -> [B, k, k]"""
        g = _mlp(rows, None, [self.conv1, self.conv2, self.conv3], [self.bn1, self.bn2, self.bn3], pool_k=N)   # :28-31
        h = _mlp(g, None, [self.fc1, self.fc2], [self.bn4, self.bn5])                                           # :34-35
        x = F.linear(h, self.fc3.weight, self.fc3.bias)                                                         # :36
        iden = torch.eye(self.k, dtype=x.dtype, device=x.device).reshape(1, self.k * self.k)                    # :38-42
        return (x + iden).view(-1, self.k, self.k)

    def forward(self, x):
        B, _, N = x.shape
        return self.forward_rows(_rows(x), B, N)


class STNkd(STN3d):                                       # reference :46-82
    def __init__(self, k=64):
        nn.Module.__init__(self)
        self.conv1 = torch.nn.Conv1d(k, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 128, 1)
        self.conv3 = torch.nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.k = k


class PointNetEncoder(nn.Module):                         # reference :85-133
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        self.stn = STN3d(channel)
        self.conv1 = torch.nn.Conv1d(channel, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 128, 1)
        self.conv3 = torch.nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if self.feature_transform:
            self.fstn = STNkd(k=64)

    def forward_rows(self, x):
        """x [B, D, N] -> (global [B, 1024], pointfeat rows [B*N, 64], trans, trans_feat)"""
        B, D, N = x.shape
        pts = x.permute(0, 2, 1).contiguous()                                    # [B, N, D]
        trans = self.stn.forward_rows(pts.view(B * N, D), B, N)                  # :101
        xyz = torch.bmm(pts[:, :, :3], trans)                                    # :106
        pts = torch.cat([xyz, pts[:, :, 3:]], dim=2) if D > 3 else xyz           # :107-108
        h = _mlp(pts.reshape(B * N, D), None, [self.conv1], [self.bn1])          # :110
        trans_feat = None
        if self.feature_transform:                                              # :112-116
            trans_feat = self.fstn.forward_rows(h, B, N)
            h = torch.bmm(h.view(B, N, 64), trans_feat).reshape(B * N, 64)
        pointfeat = h
        g = _mlp(h, None, [self.conv2], [self.bn2])                              # :121
        # bn3(conv3(x)) WITHOUT a ReLU, then the max over the points (:122-124): rocBLAS + ATen
        z = F.linear(g, self.conv3.weight.view(1024, 128), self.conv3.bias)
        z = F.batch_norm(z, self.bn3.running_mean, self.bn3.running_var, self.bn3.weight, self.bn3.bias,
                         self.bn3.training, 0.0 if self.bn3.momentum is None else self.bn3.momentum, self.bn3.eps)
        if self.bn3.training and self.bn3.num_batches_tracked is not None:
            self.bn3.num_batches_tracked.add_(1)
        return z.view(B, N, 1024).max(dim=1)[0], pointfeat, trans, trans_feat

    def forward(self, x):
        B, D, N = x.shape
        g, pointfeat, trans, trans_feat = self.forward_rows(x)
        if self.global_feat:
            return g, trans, trans_feat
        xg = g.view(B, 1024, 1).repeat(1, 1, N)                                  # :130-131
        return torch.cat([xg, pointfeat.view(B, N, 64).permute(0, 2, 1)], 1), trans, trans_feat


def feature_transform_reguliarzer(trans):                 # reference :136-142
    d = trans.size()[1]
    I = torch.eye(d, device=trans.device)[None, :, :]
    return torch.mean(torch.norm(torch.bmm(trans, trans.transpose(2, 1)) - I, dim=(1, 2)))
