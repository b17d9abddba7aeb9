"""CPU ORACLE -- test infrastructure, NOT the product path.

Python face of oracle/pn2_oracle.c (ctypes) plus a plain-torch CPU restatement of the
`pointnet2_sem_seg` network (reference models/pointnet2_sem_seg.py:6-50 on top of
models/pointnet2_utils.py:161-202, 265-315).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Parity status: PINNED against outputs of the reference itself (tests/golden/, produced by
oracle/make_golden.py which imports /root/reference on CPU); see tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpn2oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/pn2_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp, ip = ctypes.c_void_p, ctypes.c_void_p
        ci, cl = ctypes.c_int, ctypes.c_int64
        L.orc_num_threads.restype = ci
        L.orc_set_num_threads.argtypes = [ci]
        L.orc_square_distance.argtypes = [fp, fp, ci, ci, ci, fp]
        L.orc_farthest_point_sample.argtypes = [fp, ci, ci, ci, ip, ip]
        L.orc_farthest_point_sample.restype = ci
        L.orc_query_ball_point.argtypes = [ctypes.c_double, ci, fp, fp, ci, ci, ci, ip]
        L.orc_query_ball_point.restype = cl
        L.orc_index_points.argtypes = [fp, ip, ci, ci, ci, cl, fp]
        L.orc_index_points.restype = ci
        L.orc_group_points.argtypes = [fp, fp, fp, ip, ci, ci, ci, ci, ci, fp]
        L.orc_group_points.restype = ci
        L.orc_index_points_backward.argtypes = [fp, ip, ci, ci, ci, cl, ci, ci, fp]
        L.orc_three_nn.argtypes = [fp, fp, ci, ci, ci, ip, fp, fp]
        L.orc_three_nn.restype = ci
        L.orc_three_interpolate.argtypes = [fp, ip, fp, ci, ci, ci, ci, fp]
        L.orc_three_interpolate_backward.argtypes = [fp, ip, fp, ci, ci, ci, ci, fp]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------- numpy-level ops
def square_distance(src, dst):
    src, dst = _f32(src), _f32(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), np.float32)
    lib().orc_square_distance(_p(src), _p(dst), B, N, M, _p(out))
    return out


def farthest_point_sample(xyz, npoint, start):
    xyz, start = _f32(xyz), _i64(start)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), np.int64)
    rc = lib().orc_farthest_point_sample(_p(xyz), B, N, npoint, _p(start), _p(out))
    if rc != 0:
        raise IndexError("farthest_point_sample: start index out of range (rc=%d)" % rc)
    return out


def query_ball_point(radius, nsample, xyz, new_xyz, allow_empty=False):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = np.empty((B, S, nsample), np.int64)
    empty = lib().orc_query_ball_point(float(radius), nsample, _p(xyz), _p(new_xyz), B, N, S, _p(idx))
    if empty and not allow_empty:
        raise IndexError("query_ball_point: %d centroid(s) have no point within radius" % empty)
    return idx


def index_points(points, idx):
    points, idx = _f32(points), _i64(idx)
    B, N, C = points.shape
    M = int(np.prod(idx.shape[1:]))
    out = np.empty(idx.shape + (C,), np.float32)
    rc = lib().orc_index_points(_p(points), _p(idx), B, N, C, M, _p(out))
    if rc != 0:
        raise IndexError("index_points: index out of range")
    return out


def group_points(xyz, new_xyz, points, idx):
    xyz, new_xyz, idx = _f32(xyz), _f32(new_xyz), _i64(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    pts = None if points is None else _f32(points)
    out = np.empty((B, S, K, 3 + D), np.float32)
    rc = lib().orc_group_points(_p(xyz), _p(new_xyz), None if pts is None else _p(pts), _p(idx),
                                B, N, S, K, D, _p(out))
    if rc != 0:
        raise IndexError("group_points: index out of range")
    return out


def index_points_backward(grad_out, idx, N, D, col0=0):
    grad_out, idx = _f32(grad_out), _i64(idx)
    B = idx.shape[0]
    M = int(np.prod(idx.shape[1:]))
    Cg = grad_out.shape[-1]
    gp = np.zeros((B, N, D), np.float32)
    lib().orc_index_points_backward(_p(grad_out), _p(idx), B, N, D, M, Cg, col0, _p(gp))
    return gp


def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = np.empty((B, N, 3), np.int64)
    d = np.empty((B, N, 3), np.float32)
    w = np.empty((B, N, 3), np.float32)
    rc = lib().orc_three_nn(_p(xyz1), _p(xyz2), B, N, S, _p(idx), _p(d), _p(w))
    if rc != 0:
        raise ValueError("three_nn needs S >= 3")
    return idx, d, w


def three_interpolate(points2, idx3, weight3):
    points2, idx3, weight3 = _f32(points2), _i64(idx3), _f32(weight3)
    B, S, D = points2.shape
    N = idx3.shape[1]
    out = np.empty((B, N, D), np.float32)
    lib().orc_three_interpolate(_p(points2), _p(idx3), _p(weight3), B, N, S, D, _p(out))
    return out


def three_interpolate_backward(grad_out, idx3, weight3, S):
    grad_out, idx3, weight3 = _f32(grad_out), _i64(idx3), _f32(weight3)
    B, N, D = grad_out.shape
    g = np.zeros((B, S, D), np.float32)
    lib().orc_three_interpolate_backward(_p(grad_out), _p(idx3), _p(weight3), B, N, S, D, _p(g))
    return g


# ----------------------------------------------------------------------------- network restatement
SA_CFG = (  # reference models/pointnet2_sem_seg.py:9-12
    ("sa1", 1024, 0.1, 32, (32, 32, 64)),
    ("sa2", 256, 0.2, 32, (64, 64, 128)),
    ("sa3", 64, 0.4, 32, (128, 128, 256)),
    ("sa4", 16, 0.8, 32, (256, 256, 512)),
)
FP_CFG = (  # reference models/pointnet2_sem_seg.py:13-16
    ("fp4", 768, (256, 256)),
    ("fp3", 384, (256, 256)),
    ("fp2", 320, (256, 128)),
    ("fp1", 128, (128, 128, 128)),
)


def state_shapes(num_classes=18, num_extra_features=3):
    """{state_dict key: shape} in the reference's registration order."""
    shapes = {}

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    cin = 6 + 3 + num_extra_features
    for name, _, _, _, mlp in SA_CFG:
        last = cin
        for i, co in enumerate(mlp):
            shapes["%s.mlp_convs.%d.weight" % (name, i)] = (co, last, 1, 1)
            shapes["%s.mlp_convs.%d.bias" % (name, i)] = (co,)
            last = co
        for i, co in enumerate(mlp):
            bn("%s.mlp_bns.%d" % (name, i), co)
        cin = mlp[-1] + 3
    for name, cin, mlp in FP_CFG:
        last = cin
        for i, co in enumerate(mlp):
            shapes["%s.mlp_convs.%d.weight" % (name, i)] = (co, last, 1)
            shapes["%s.mlp_convs.%d.bias" % (name, i)] = (co,)
            last = co
        for i, co in enumerate(mlp):
            bn("%s.mlp_bns.%d" % (name, i), co)
    shapes["conv1.weight"] = (128, 128, 1)
    shapes["conv1.bias"] = (128,)
    bn("bn1", 128)
    shapes["conv2.weight"] = (num_classes, 128, 1)
    shapes["conv2.bias"] = (num_classes,)
    return shapes


class OracleNet:
    """Functional torch-CPU restatement; parameters live in `self.sd` (name -> tensor)."""

    def __init__(self, state, bn_momentum=0.1, dropout_p=0.5, dtype=None):
        """dtype=torch.float64: the same network evaluated in double precision on the SAME fp32 geometry (FPS /
        ball-query / 3-NN indices and interpolation weights are computed from the fp32 coordinates as always): the
        yardstick for how far any fp32 evaluation order may sit from the exact gradients."""
        import torch
        self.torch = torch
        self.dtype = dtype or torch.float32
        self.sd = {}
        for k, v in state.items():
            t = torch.as_tensor(np.asarray(v)).clone()
            if t.is_floating_point():
                t = t.to(self.dtype)
            if t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                t.requires_grad_(True)
            self.sd[k] = t
        self.bn_momentum = bn_momentum
        self.dropout_p = dropout_p
        self.training = False
        self.taps = {}
        # gate pinning (tests/test_hip_gates.py): {"sa1": {"relu": [mask per layer, oracle layout], "pool": winning k
        # [B,C,1,S] int64}, ..., "fp1": {"relu": [...]}, "head": {"relu": [mask]}}.  A pinned ReLU is `x * mask`, a pinned
        # max-pool a gather: the network becomes the piecewise-linear branch somebody else's decisions selected.
        self.gates = None

    def parameters(self):
        return [v for v in self.sd.values() if v.requires_grad]

    def named_parameters(self):
        return [(k, v) for k, v in self.sd.items() if v.requires_grad]

    def _conv_bn_relu(self, x, prefix, i):
        F = self.torch.nn.functional
        w = self.sd["%s.mlp_convs.%d.weight" % (prefix, i)]
        b = self.sd["%s.mlp_convs.%d.bias" % (prefix, i)]
        x = F.conv2d(x, w, b) if w.dim() == 4 else F.conv1d(x, w, b)
        x = self._bn(x, "%s.mlp_bns.%d" % (prefix, i))
        if self.gates is not None and prefix in self.gates:
            return x * self.gates[prefix]["relu"][i].to(x.dtype)
        return F.relu(x)

    def _bn(self, x, p):
        F = self.torch.nn.functional
        if self.training:
            self.sd[p + ".num_batches_tracked"] += 1
        return F.batch_norm(x, self.sd[p + ".running_mean"], self.sd[p + ".running_var"],
                            self.sd[p + ".weight"], self.sd[p + ".bias"],
                            self.training, self.bn_momentum, 1e-5)

    def set_abstraction(self, name, npoint, radius, nsample, nlayers, xyz, points, start):
        """xyz [B,N,3], points [B,N,D] -> new_xyz [B,S,3], feats [B,S,C'] (pointnet2_utils.py:176-202)."""
        torch = self.torch
        xyz_np = xyz.detach().numpy()
        fps = farthest_point_sample(xyz_np, npoint, start)
        new_xyz_np = index_points(xyz_np, fps)
        idx = query_ball_point(radius, nsample, xyz_np, new_xyz_np)
        self.taps[name + ".fps_idx"] = fps
        self.taps[name + ".ball_idx"] = idx
        tidx = torch.from_numpy(idx)
        bsel = torch.arange(xyz.shape[0]).view(-1, 1, 1)
        new_xyz = torch.from_numpy(new_xyz_np).to(self.dtype)
        g_xyz = xyz[bsel, tidx] - new_xyz.unsqueeze(2)
        grouped = torch.cat([g_xyz, points[bsel, tidx]], dim=-1)          # [B,S,K,3+D]
        x = grouped.permute(0, 3, 2, 1)                                   # [B,C,K,S]
        for i in range(nlayers):
            x = self._conv_bn_relu(x, name, i)
        if self.gates is not None and name in self.gates and self.gates[name].get("pool") is not None:
            x = x.gather(2, self.gates[name]["pool"]).squeeze(2)          # the pinned winner of every group
        else:
            x = x.max(dim=2)[0]                                           # [B,C',S]
        return new_xyz, x.permute(0, 2, 1)

    def feature_propagation(self, name, nlayers, xyz1, xyz2, points1, points2):
        """xyz1 [B,N,3], xyz2 [B,S,3], points1 [B,N,D1]|None, points2 [B,S,D2] -> [B,N,C']
        (pointnet2_utils.py:276-315)."""
        torch = self.torch
        idx3, _, w3 = three_nn(xyz1.detach().numpy(), xyz2.detach().numpy())
        self.taps[name + ".nn_idx"] = idx3
        self.taps[name + ".nn_weight"] = w3
        tidx, tw = torch.from_numpy(idx3), torch.from_numpy(w3).to(self.dtype)
        bsel = torch.arange(xyz1.shape[0]).view(-1, 1, 1)
        interp = (points2[bsel, tidx] * tw.unsqueeze(-1)).sum(dim=2)      # [B,N,D2]
        x = interp if points1 is None else torch.cat([points1, interp], dim=-1)
        x = x.permute(0, 2, 1)
        for i in range(nlayers):
            x = self._conv_bn_relu(x, name, i)
        return x.permute(0, 2, 1)

    def forward(self, blocks, fps_starts):
        """blocks [B,C,N] channel-first like the reference (pointnet2_sem_seg.py:22-40);
        fps_starts = 4 arrays [B].  Returns log-probs [B,N,classes] and l4 feats [B,512,16]."""
        torch = self.torch
        F = torch.nn.functional
        x = torch.as_tensor(blocks, dtype=self.dtype)
        pts = x.permute(0, 2, 1).contiguous()                              # [B,N,C]
        xyzs, feats = [pts[:, :, :3].contiguous()], [pts]
        for (name, npoint, radius, nsample, mlp), st in zip(SA_CFG, fps_starts):
            nx, nf = self.set_abstraction(name, npoint, radius, nsample, len(mlp), xyzs[-1], feats[-1], st)
            xyzs.append(nx)
            feats.append(nf)
            self.taps[name + ".out"] = nf
        f = feats[4]
        for lvl, (name, _, mlp) in zip((3, 2, 1, 0), FP_CFG):
            skip = feats[lvl] if lvl > 0 else None
            f = self.feature_propagation(name, len(mlp), xyzs[lvl], xyzs[lvl + 1], skip, f)
            self.taps[name + ".out"] = f
        h = f.permute(0, 2, 1)
        h = self._bn(F.conv1d(h, self.sd["conv1.weight"], self.sd["conv1.bias"]), "bn1")
        h = h * self.gates["head"]["relu"][0].to(h.dtype) if (self.gates is not None and "head" in self.gates) else F.relu(h)
        h = F.dropout(h, self.dropout_p, self.training)
        h = F.conv1d(h, self.sd["conv2.weight"], self.sd["conv2.bias"])
        logp = F.log_softmax(h, dim=1).permute(0, 2, 1)
        return logp, feats[4].permute(0, 2, 1)

    def loss(self, logp, target, weight=None):
        """get_loss (pointnet2_sem_seg.py:44-50) on the flattened view used by
        localfunctions.py:212-216."""
        torch = self.torch
        F = torch.nn.functional
        t = torch.as_tensor(target, dtype=torch.int64).reshape(-1)
        w = None if weight is None else torch.as_tensor(weight, dtype=self.dtype)
        return F.nll_loss(logp.reshape(-1, logp.shape[-1]), t, weight=w)

    def train_step(self, blocks, target, fps_starts, optimizer, weight=None):
        """zero_grad -> forward -> nll_loss -> backward -> step (localfunctions.py:203-218)."""
        self.training = True
        optimizer.zero_grad()
        logp, _ = self.forward(blocks, fps_starts)
        loss = self.loss(logp, target, weight)
        loss.backward()
        optimizer.step()
        return float(loss.detach())


def make_adam(params, lr=1e-3, weight_decay=1e-4):
    """Optimizer of the reference driver: sem_seg_training.py:576-582."""
    import torch
    return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
