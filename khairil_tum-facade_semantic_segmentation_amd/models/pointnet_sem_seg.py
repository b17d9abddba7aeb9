"""pointnet_sem_seg -- the plain-PointNet control model of BASELINE configs[4] (reference
models/pointnet_sem_seg.py:9-47): same constructor, forward signature ([B, C, N] -> log-probabilities [B, N, classes],
trans_feat) and state_dict keys.  The segmentation head's conv/BN/ReLU chain reads its 1088-channel input from two
sources ([global feature | point feature]) without the concatenated tensor, and conv4 + log_softmax is the same
head kernel pointnet2_sem_seg uses."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import head
from . import pointnet2_utils as _utils
from .pointnet_utils import PointNetEncoder, feature_transform_reguliarzer


class get_model(nn.Module):                               # reference :9-34
    def __init__(self, num_class, num_extra_features):
        super().__init__()
        self.k = num_class
        num_of_channels = 6 + num_extra_features
        self.feat = PointNetEncoder(global_feat=False, feature_transform=True, channel=num_of_channels)
        self.conv1 = torch.nn.Conv1d(1088, 512, 1)
        self.conv2 = torch.nn.Conv1d(512, 256, 1)
        self.conv3 = torch.nn.Conv1d(256, 128, 1)
        self.conv4 = torch.nn.Conv1d(128, self.k, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(128)

    def forward(self, x):
        B, _, N = x.shape
        g, pointfeat, _, trans_feat = self.feat.forward_rows(x)                 # :25
        xg = g.view(B, 1, 1024).expand(B, N, 1024).reshape(B * N, 1024)          # the repeat of pointnet_utils.py:130
        h = _utils._mlp(xg, pointfeat, [self.conv1, self.conv2, self.conv3], [self.bn1, self.bn2, self.bn3])   # :26-28
        if _utils._TORCH_MLP or self.k > 32:
            logp = F.log_softmax(F.linear(h, self.conv4.weight.view(self.k, 128), self.conv4.bias), dim=-1)
        else:
            logp = head.head_logits(h, self.conv4.weight, self.conv4.bias)       # conv4 -> log_softmax (:29-31)
        return logp.view(B, N, self.k), trans_feat


class get_loss(torch.nn.Module):                          # reference :36-46
    def __init__(self, mat_diff_loss_scale=0.001):
        super().__init__()
        self.mat_diff_loss_scale = mat_diff_loss_scale

    def forward(self, pred, target, trans_feat, weight):
        loss = F.nll_loss(pred, target, weight=weight) if _utils._TORCH_MLP else head.nll_loss(pred, target, weight)
        return loss + feature_transform_reguliarzer(trans_feat) * self.mat_diff_loss_scale


# multiply-accumulates per input point of one forward pass (~1.146 MMAC): for bench.py's TFLOP/s
def macs_per_point(num_channels, num_class):
    stn = num_channels * 64 + 64 * 128 + 128 * 1024
    fstn = 64 * 64 + 64 * 128 + 128 * 1024
    enc = 3 * 3 + num_channels * 64 + 64 * 64 + 64 * 128 + 128 * 1024
    headm = 1088 * 512 + 512 * 256 + 256 * 128 + 128 * num_class
    return stn + fstn + enc + headm
