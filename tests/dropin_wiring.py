"""The CALLER of the drop-in operator surface, restated (tests and bench.py --drop-in only; not product code).

The reference's models/pointnet2_sem_seg.py:6-50 imports PointNetSetAbstraction / PointNetFeaturePropagation from
models.pointnet2_utils and wires them channel-first with a torch head (Conv1d -> BatchNorm1d -> ReLU -> Dropout ->
Conv1d -> log_softmax) and F.nll_loss.  A maintainer who swaps only pointnet2_utils.py for this package's module runs
exactly this: none of the package's own fast-path wiring (channel-last pyramid, geometry prefetch, hipGraph, fused
head, device Adam) is involved.  Same submodule names, so the reference's state_dict loads."""
import torch.nn as nn
import torch.nn.functional as F


def build(utils, num_classes, num_extra_features):
    """utils = the module that plays models.pointnet2_utils (the package's drop-in)."""
    SA, FP = utils.PointNetSetAbstraction, utils.PointNetFeaturePropagation

    class ReferenceWiredModel(nn.Module):
        def __init__(self):
            super().__init__()
            widths = [(1024, 0.1, 6 + 3 + num_extra_features, [32, 32, 64]), (256, 0.2, 64 + 3, [64, 64, 128]),
                      (64, 0.4, 128 + 3, [128, 128, 256]), (16, 0.8, 256 + 3, [256, 256, 512])]
            for i, (npoint, radius, cin, mlp) in enumerate(widths, start=1):
                setattr(self, "sa%d" % i, SA(npoint, radius, 32, cin, mlp, False))
            for name, cin, mlp in (("fp4", 768, [256, 256]), ("fp3", 384, [256, 256]), ("fp2", 320, [256, 128]),
                                   ("fp1", 128, [128, 128, 128])):
                setattr(self, name, FP(cin, mlp))
            self.conv1 = nn.Conv1d(128, 128, 1)
            self.bn1 = nn.BatchNorm1d(128)
            self.drop1 = nn.Dropout(0.5)
            self.conv2 = nn.Conv1d(128, num_classes, 1)

        def forward(self, xyz):                                  # [B, 6 + extra, N], channel-first throughout
            pts = [xyz]
            pos = [xyz[:, :3, :]]
            for sa in (self.sa1, self.sa2, self.sa3, self.sa4):
                p, f = sa(pos[-1], pts[-1])
                pos.append(p)
                pts.append(f)
            up = pts[4]
            for lvl, fp in zip((3, 2, 1, 0), (self.fp4, self.fp3, self.fp2, self.fp1)):
                up = fp(pos[lvl], pos[lvl + 1], pts[lvl] if lvl else None, up)
            x = self.drop1(F.relu(self.bn1(self.conv1(up))))
            x = F.log_softmax(self.conv2(x), dim=1)
            return x.permute(0, 2, 1), pts[4]

    return ReferenceWiredModel()


def loss_fn(pred, target, weight):
    return F.nll_loss(pred, target, weight=weight)
