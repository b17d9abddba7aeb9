"""Synthetic 4096-point blocks.  numpy only: the same inputs are regenerated on
the GPU box, in the oracle and in oracle/make_golden.py from a seed, never from torch's RNG.

Block layout follows TrainCustomDataset.__getitem__ (reference sem_seg_training.py:224-253):
cols 0:3 = xyz (x,y centred on the block, z raw), cols 3:6 = xyz / room_max in [0,1],
cols 6:9 = RGB/255 (absent with --RGB_OFF)."""
import numpy as np

BENCH_SEED = 20231003          # + rank
WEIGHT_SEED = 7


def make_xyz(rs, B, N, kind="cube"):
    """xyz[B,N,3] float32.  cube: x,y~U(-.5,.5), z~U(0,1) (mean ~14.8 hits at r=0.1);
    facade: x~U(-.5,.5), y~N(0,.02), z~U(0,3) (dense slab, ~71% of SA1 balls reach 32 hits)."""
    if kind == "cube":
        x = rs.uniform(-0.5, 0.5, size=(B, N))
        y = rs.uniform(-0.5, 0.5, size=(B, N))
        z = rs.uniform(0.0, 1.0, size=(B, N))
    elif kind == "facade":
        x = rs.uniform(-0.5, 0.5, size=(B, N))
        y = rs.normal(0.0, 0.02, size=(B, N))
        z = rs.uniform(0.0, 3.0, size=(B, N))
    else:
        raise ValueError("unknown synthetic distribution %r" % (kind,))
    return np.stack([x, y, z], axis=-1).astype(np.float32)


def make_blocks(rs, B, N, C=9, kind="cube"):
    """Network input blocks [B,N,C] float32 (C = 9 with RGB, 6 with --RGB_OFF)."""
    if C < 3:
        raise ValueError("C must be >= 3")
    xyz = make_xyz(rs, B, N, kind)
    extra = rs.uniform(0.0, 1.0, size=(B, N, C - 3)).astype(np.float32)
    return np.concatenate([xyz, extra], axis=-1)


def make_labels(rs, B, N, num_classes=18):
    return rs.randint(0, num_classes, size=(B, N)).astype(np.int64)


def make_fps_starts(rs, B, sizes=(4096, 1024, 256, 64)):
    """One injected FPS start index per (level, block): the reference draws them with
    torch.randint inside farthest_point_sample (models/pointnet2_utils.py:75)."""
    return [rs.randint(0, n, size=(B,)).astype(np.int64) for n in sizes]


def draw_case(seed, B, N, C=9, kind="cube", num_classes=18):
    """One seeded test/bench case, always drawn in this order: blocks, labels, the four FPS start
    vectors, class weights.  Golden fixtures store only `seed` (and the starts) for the inputs."""
    rs = np.random.RandomState(seed)
    blocks = make_blocks(rs, B, N, C, kind)
    labels = make_labels(rs, B, N, num_classes)
    starts = make_fps_starts(rs, B, sizes=(N, 1024, 256, 64))
    class_weight = rs.uniform(0.5, 1.5, size=(num_classes,)).astype(np.float32)
    return blocks, labels, starts, class_weight


def fill_state_dict(shapes, seed=WEIGHT_SEED):
    """Deterministic parameter/buffer values for a {name: shape} dict (insertion ordered).
    conv weights: xavier-scaled normal; biases and BN beta small; BN gamma in [.5,1.5];
    running stats non-trivial so eval-mode BN is exercised."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = rs.normal(0.0, 0.1, size=shape).astype(np.float32)
        elif leaf == "running_var":
            out[name] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif leaf == "weight" and len(shape) >= 2:
            fan_out, fan_in = shape[0], shape[1]
            std = np.sqrt(2.0 / (fan_in + fan_out))
            out[name] = rs.normal(0.0, std, size=shape).astype(np.float32)
        elif leaf == "weight":
            out[name] = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
        else:  # bias
            out[name] = rs.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    return out
