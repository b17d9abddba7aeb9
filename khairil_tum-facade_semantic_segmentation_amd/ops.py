"""Host-side operator layer: torch tensors in, libpn2hip.so (C ABI) underneath.

torch is plumbing here (device memory, streams, autograd bookkeeping); every function below
enqueues hand-written HIP kernels on torch's current stream through include/pn2_hip.h.  There is
deliberately no CPU implementation: CPU tensors raise."""
import torch

from . import _lib

import os

_ERR = {}
_ERROR_MODE = "lazy"
_SKIP_INPLACE = os.environ.get("PN2_SKIP_INPLACE", "0") == "1"


def set_error_mode(mode):
    """'eager': synchronise and raise IndexError right after an op that can fault (the
    reference's behaviour, models/pointnet2_utils.py:59).  'lazy' (default): faults are
    counted on the device and raised by check_errors()."""
    global _ERROR_MODE
    if mode not in ("lazy", "eager"):
        raise ValueError(mode)
    _ERROR_MODE = mode


def _err_word(device):
    key = (device.type, device.index)
    t = _ERR.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _ERR[key] = t
    return t


def check_errors(device=None, what="pn2 op"):
    """Raise IndexError if any kernel since the last check saw an empty ball / bad index."""
    for key, t in list(_ERR.items()):
        if device is not None and (device.type, device.index) != key:
            continue
        n = int(t.item())
        if n:
            t.zero_()
            raise IndexError("%s: %d out-of-range index / empty ball-query neighbourhood(s) "
                             "(the reference raises IndexError in index_points)" % (what, n))


def _after_fault_op(device, what):
    if _ERROR_MODE == "eager":
        check_errors(device, what)


def _dev(*tensors):
    d = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pn2 operators run on the HIP device only (got a %s tensor); there is "
                               "no CPU fallback in the product path" % t.device)
        if d is None:
            d = t.device
        elif t.device != d:
            raise RuntimeError("tensors on different devices: %s vs %s" % (d, t.device))
    return d


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _i64c(t):
    return t.detach().to(torch.int64).contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------- sampling
def farthest_point_sample_with_xyz(xyz, npoint, start=None):
    """xyz [B,N,3] -> (idx [B,npoint] int64, new_xyz [B,npoint,3]).  `start` [B] are the start
    indices; None draws them like the reference does (models/pointnet2_utils.py:75)."""
    dev = _dev(xyz, start)
    lib = _lib.load()
    xyz = _f32c(xyz)
    B, N, C = xyz.shape
    if C != 3:
        raise ValueError("xyz must be [B,N,3]")
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=dev)
    start = _i64c(start)
    if start.shape != (B,):
        raise ValueError("start must be [B]")
    idx = torch.empty((B, npoint), dtype=torch.int64, device=dev)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_farthest_point_sample(_ptr(xyz), B, N, npoint, _ptr(start), _ptr(idx), _ptr(new_xyz),
                                           _ptr(_err_word(dev)), _stream(dev))
    _lib.check(rc, "pn2_farthest_point_sample")
    _after_fault_op(dev, "farthest_point_sample")
    return idx, new_xyz


def farthest_point_sample(xyz, npoint, start=None):
    return farthest_point_sample_with_xyz(xyz, npoint, start)[0]


class BallPlan:
    """Per-block query plans of the binned ball query (csrc/pn2_ball_bin.h): cell-sorted points, per-centroid
    candidate runs and, once pack_rows() has run, the packed rows the grouping gathers from.  Built by
    farthest_point_sample_plan() (tail of the FPS kernel) or ball_plan() (stand-alone kernel); valid for one
    (xyz, new_xyz, radius) triple."""

    def __init__(self, buf, B, N, S, D, radius, xyz=None, new_xyz=None):
        self.buf, self.B, self.N, self.S, self.D, self.radius = buf, B, N, S, D, float(radius)
        self.rows_packed = False
        # the tensors the plan was built from (held: their storage cannot be recycled under the plan) and, once rows are
        # packed, the features they were packed from
        self.xyz, self.new_xyz, self.points = xyz, new_xyz, None

    @staticmethod
    def _same(held, t):
        return held is None or t is None or held is t or (held.data_ptr() == t.data_ptr() and held.shape == t.shape
                                                          and held._version == t._version)

    def matches(self, B, N, S, D, radius, xyz=None, new_xyz=None):
        """True when the plan was built for this geometry: same sizes and radius, and (when the tensors are given) the
        same xyz / new_xyz tensors -- a plan of another cloud of the same shape would silently return that cloud's
        neighbours."""
        return ((self.B, self.N, self.S, self.D) == (B, N, S, D) and self.radius == float(radius)
                and self._same(self.xyz, xyz) and self._same(self.new_xyz, new_xyz))

    def pack_rows(self, xyz, points):
        if self.rows_packed and self._same(self.points, points):
            return
        lib = _lib.load()
        dev = self.buf.device
        with torch.cuda.device(dev):
            rc = lib.pn2_ball_pack_rows(_ptr(xyz), _ptr(points), self.B, self.N, self.S, self.D, _ptr(self.buf), _stream(dev))
        _lib.check(rc, "pn2_ball_pack_rows")
        self.rows_packed = True
        self.points = points


def plan_supported(B, N, S):
    """Shapes the planned ball query is built for: many centroids over blocks of a few thousand points."""
    return B * S >= 4096 and 1024 < N <= 8192


def _plan_buffer(B, N, S, D, dev):
    nbytes = _lib.load().pn2_ball_plan_bytes(N, S, D)
    if nbytes <= 0:
        raise RuntimeError("pn2_ball_plan_bytes(%d, %d, %d): unsupported shape" % (N, S, D))
    return torch.empty((B, nbytes), dtype=torch.uint8, device=dev)       # torch allocations are >= 256-byte aligned


def farthest_point_sample_plan(xyz, npoint, radius, D, start=None):
    """farthest_point_sample_with_xyz whose kernel also leaves the ball-query plan of every block for `radius`
    (D = feature width of the rows that will be grouped).  -> (idx, new_xyz, BallPlan)."""
    dev = _dev(xyz, start)
    lib = _lib.load()
    xyz = _f32c(xyz)
    B, N, C = xyz.shape
    if C != 3:
        raise ValueError("xyz must be [B,N,3]")
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=dev)
    start = _i64c(start)
    if start.shape != (B,):
        raise ValueError("start must be [B]")
    idx = torch.empty((B, npoint), dtype=torch.int64, device=dev)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=dev)
    buf = _plan_buffer(B, N, npoint, D, dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_farthest_point_sample_plan(_ptr(xyz), B, N, npoint, _ptr(start), _ptr(idx), _ptr(new_xyz), float(radius),
                                                int(D), _ptr(buf), _ptr(_err_word(dev)), _stream(dev))
    _lib.check(rc, "pn2_farthest_point_sample_plan")
    _after_fault_op(dev, "farthest_point_sample")
    return idx, new_xyz, BallPlan(buf, B, N, npoint, D, radius, xyz, new_xyz)


def ball_plan(radius, xyz, new_xyz, points=None):
    """Stand-alone plan (cell sort + per-centroid runs + packed rows) for query_ball_point / ball_query_group."""
    dev = _dev(xyz, new_xyz, points)
    lib = _lib.load()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    D = 0 if points is None else points.shape[2]
    buf = _plan_buffer(B, N, S, D, dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_ball_plan(float(radius), _ptr(xyz), _ptr(new_xyz), _ptr(points), B, N, S, D, _ptr(buf), _stream(dev))
    _lib.check(rc, "pn2_ball_plan")
    plan = BallPlan(buf, B, N, S, D, radius, xyz, new_xyz)
    plan.rows_packed = True
    plan.points = points
    return plan


def square_distance(src, dst):
    dev = _dev(src, dst)
    lib = _lib.load()
    src, dst = _f32c(src), _f32c(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty((B, N, M), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_square_distance(_ptr(src), _ptr(dst), B, N, M, _ptr(out), _stream(dev))
    _lib.check(rc, "pn2_square_distance")
    return out


# ---------------------------------------------------------------------------------- grouping
def _padded(width, pad_to):
    return (width + pad_to - 1) // pad_to * pad_to


_NEXT_GROUPED_OUT = None


def place_next_grouped(out):
    """The NEXT grouped-rows result of ball_query_group goes into `out` ([B,S,K,ldg] fp32, contiguous) instead of a fresh
    tensor, if the shapes agree (otherwise the offer lapses).  The trainer places the first level's rows -- 25 MB of the
    27 MB geometry pyramid -- directly in the buffer the next step reads them from."""
    global _NEXT_GROUPED_OUT
    _NEXT_GROUPED_OUT = out


def _ball_query_group_raw(radius, nsample, xyz, new_xyz, points, want_grouped, pad_to=1, plan=None):
    dev = _dev(xyz, new_xyz, points)
    lib = _lib.load()
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    D = 0 if points is None else points.shape[2]
    ldg = _padded(3 + D, pad_to)
    idx = torch.empty((B, S, nsample), dtype=torch.int64, device=dev)
    grouped = None
    if want_grouped:
        global _NEXT_GROUPED_OUT
        placed, _NEXT_GROUPED_OUT = _NEXT_GROUPED_OUT, None
        if placed is not None and tuple(placed.shape) == (B, S, nsample, ldg) and placed.dtype == torch.float32 \
                and placed.device == dev and placed.is_contiguous():
            grouped = placed                     # the caller's buffer (place_next_grouped): no copy of the rows later
        else:
            grouped = torch.empty((B, S, nsample, ldg), dtype=torch.float32, device=dev)
    if plan is not None and not (plan.matches(B, N, S, plan.D, radius, xyz, new_xyz) and (plan.D == D or not want_grouped)):
        plan = None
    # No plan given: the self-contained entry (one launch).  Building a plan for ONE query costs more than it saves
    # (SA1, B = 16: plan 11.5 us + planned query 8.9 us against 19.3 us self-contained) -- a plan pays
    # for callers that reuse it, who build it with ball_plan() / farthest_point_sample_plan() and pass it in.
    if plan is not None and want_grouped and "q1" in _LAB_SKIP:
        # lab switch (wrong results): the planned query of a shape is launched once, later calls return its indices and leave
        # the rows buffer as it is -- what the row packing + query launches cost the step beside them
        key = ("q1", B, N, S, nsample, ldg)
        if key in _LAB_CACHE:
            return _LAB_CACHE[key], grouped
    if plan is not None:
        if want_grouped and ldg == 3 + D and (3 + D) % 4 == 0:
            plan.pack_rows(xyz, points)                       # the fused row stores gather from the plan's packed rows
        with torch.cuda.device(dev):
            rc = lib.pn2_ball_query_group_planned(float(radius), int(nsample), _ptr(plan.buf), _ptr(xyz), _ptr(new_xyz),
                                                  _ptr(points), B, N, S, plan.D, _ptr(idx), _ptr(grouped), ldg,
                                                  _ptr(_err_word(dev)), _stream(dev))
        _lib.check(rc, "pn2_ball_query_group_planned")
    else:
        with torch.cuda.device(dev):
            rc = lib.pn2_ball_query_group(float(radius), int(nsample), _ptr(xyz), _ptr(new_xyz), _ptr(points), B, N, S,
                                          D, _ptr(idx), _ptr(grouped), ldg, _ptr(_err_word(dev)), _stream(dev))
        _lib.check(rc, "pn2_ball_query_group")
    _after_fault_op(dev, "query_ball_point")
    if plan is not None and want_grouped and "q1" in _LAB_SKIP:
        _LAB_CACHE[("q1", B, N, S, nsample, ldg)] = idx
    return idx, grouped


def query_ball_point(radius, nsample, xyz, new_xyz, plan=None):
    idx, _ = _ball_query_group_raw(radius, nsample, _f32c(xyz), _f32c(new_xyz), None, False, plan=plan)
    return idx


class _BallQueryGroup(torch.autograd.Function):
    """idx, grouped = f(xyz, new_xyz, points); d grouped / d points is a scatter-add.
    xyz / new_xyz receive no gradient: in the network they derive from the input cloud only
    (SURVEY.md 3.3)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, points, radius, nsample, pad_to, plan=None):
        idx, grouped = _ball_query_group_raw(radius, nsample, xyz, new_xyz, points, True, pad_to, plan=plan)
        ctx.save_for_backward(idx)
        ctx.shape = (xyz.shape[0], xyz.shape[1], 0 if points is None else points.shape[2])
        ctx.mark_non_differentiable(idx)
        return idx, grouped

    @staticmethod
    def backward(ctx, _gidx, ggrouped):
        (idx,) = ctx.saved_tensors
        B, N, D = ctx.shape
        if D == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None, None, None
        return None, None, index_points_backward(ggrouped, idx, N, D, col0=3), None, None, None, None


def ball_query_group(radius, nsample, xyz, new_xyz, points, pad_to=1, plan=None):
    """Fused query_ball_point + grouping: (idx [B,S,K] int64, grouped [B,S,K,3+D]).  pad_to = 4
    rounds the row width up to a multiple of 4 floats (extra columns are zero) so that the MLP
    kernels can use 16-byte loads on widths like 67 / 131 / 259.  plan = the BallPlan that
    farthest_point_sample_plan() left for these (xyz, new_xyz, radius)."""
    xyz, new_xyz = _f32c(xyz), _f32c(new_xyz)
    if points is not None:
        points = points.to(torch.float32).contiguous()
    return _BallQueryGroup.apply(xyz, new_xyz, points, radius, nsample, pad_to, plan)


def invert_index(idx, nkeys):
    """Transposed index table of idx [B, ...] with values in [0, nkeys): (offsets [B,nkeys+1] int32,
    entries [B,E] int32) for the gather-sum form of the backward passes, or None when the table is too
    large for the on-chip transposition (the scatter-add operators are used then)."""
    if "inv" in _LAB_SKIP and not getattr(invert_index, "_inside", False):
        invert_index._inside = True
        try:
            return lab_cached("inv", (tuple(idx.shape), int(nkeys)), lambda: invert_index(idx, nkeys))
        finally:
            invert_index._inside = False
    dev = _dev(idx)
    lib = _lib.load()
    idx = _i64c(idx)
    B = idx.shape[0]
    E = idx.numel() // max(B, 1)
    off = torch.empty((B, nkeys + 1), dtype=torch.int32, device=dev)
    ent = torch.empty((B, E), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_invert_index(_ptr(idx), B, E, nkeys, _ptr(off), _ptr(ent), _stream(dev))
    if rc == -3:                                          # PN2_ERR_UNSUPPORTED
        return None
    _lib.check(rc, "pn2_invert_index")
    return off, ent


def invert_index_many(idxs, nkeys):
    """invert_index of several index tensors of one batch size in ONE launch: a list of (offsets, entries) pairs, or None
    when a table is too large for the on-chip transposition (the caller then inverts one by one / keeps the scatter-adds)."""
    import ctypes
    if not idxs:
        return []
    dev = _dev(*idxs)
    lib = _lib.load()
    idxs = [_i64c(t) for t in idxs]
    B = idxs[0].shape[0]
    if any(t.shape[0] != B for t in idxs) or len(idxs) > 8:
        return None
    Es = [t.numel() // max(B, 1) for t in idxs]
    offs = [torch.empty((B, k + 1), dtype=torch.int32, device=dev) for k in nkeys]
    ents = [torch.empty((B, e), dtype=torch.int32, device=dev) for e in Es]
    n = len(idxs)
    vp = ctypes.c_void_p * n
    with torch.cuda.device(dev):
        rc = lib.pn2_invert_index_many(n, vp(*[t.data_ptr() for t in idxs]), B, (ctypes.c_longlong * n)(*Es),
                                       (ctypes.c_int * n)(*[int(k) for k in nkeys]), vp(*[t.data_ptr() for t in offs]),
                                       vp(*[t.data_ptr() for t in ents]), _stream(dev))
    if rc == -3:                                          # PN2_ERR_UNSUPPORTED
        return None
    _lib.check(rc, "pn2_invert_index_many")
    return list(zip(offs, ents))


def _gather_sum(src, rows_src, col0, inv, weight, ediv, nkeys, D, addend=None):
    dev = _dev(src)
    lib = _lib.load()
    off, ent = inv
    B, E = ent.shape
    out = torch.empty((B, nkeys, D), dtype=torch.float32, device=dev)
    if addend is not None:
        addend = _f32c(addend)
        if tuple(addend.shape) != (B, nkeys, D):
            raise ValueError("addend %s, expected %s" % (tuple(addend.shape), (B, nkeys, D)))
    with torch.cuda.device(dev):
        rc = lib.pn2_gather_sum_add(_ptr(src), rows_src, src.shape[-1], col0, _ptr(off), _ptr(ent), _ptr(weight), E, ediv, B,
                                    nkeys, D, _ptr(addend), _ptr(out), _stream(dev))
    _lib.check(rc, "pn2_gather_sum_add")
    return out


def index_points_backward(grad_out, idx, N, D, col0=0, inv=None, into=None, into_owned=False):
    """grad_points[B,N,D] = scatter-add of grad_out[B,...,Cg][..., col0:col0+D] at idx; with
    inv = invert_index(idx, N) the same sum as an atomic-free gather in a fixed order.
    into = a contiguous fp32 [B,N,D] tensor that already holds another gradient of the same points: the scatter then
    accumulates onto a copy of it (no zero fill, no add afterwards) and returns that; into_owned = the caller vouches
    that nobody else reads `into` (see group_points(with_skip="inplace")): no copy either."""
    dev = _dev(grad_out, idx)
    lib = _lib.load()
    grad_out, idx = _f32c(grad_out), _i64c(idx)
    B = idx.shape[0]
    M = idx.numel() // max(B, 1)
    Cg = grad_out.shape[-1]
    if inv is not None:
        return _gather_sum(grad_out, M, col0, inv, None, 1, N, D, addend=into)      # the skip gradient joins the sum
    if into is not None and into.dtype == torch.float32 and into.is_contiguous() and tuple(into.shape) == (B, N, D):
        # accumulate on top of the other gradient: a COPY of it (autograd owns `into` -- it may be the same tensor
        # another node receives -- so it is never modified in place; PN2_SKIP_INPLACE=1 restores the in-place form,
        # valid for the network's own wiring where the skip gradient has a single consumer: 15 us per step)
        gp = into if (_SKIP_INPLACE or into_owned) else into.clone()
    else:
        gp = torch.zeros((B, N, D), dtype=torch.float32, device=dev)
        if into is not None:
            gp += into
    with torch.cuda.device(dev):
        rc = lib.pn2_index_points_backward(_ptr(grad_out), _ptr(idx), B, N, D, M, Cg, col0, _ptr(gp), _stream(dev))
    _lib.check(rc, "pn2_index_points_backward")
    return gp


class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        dev = _dev(points, idx)
        lib = _lib.load()
        B, N, C = points.shape
        M = idx.numel() // max(B, 1)
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pn2_index_points(_ptr(points), _ptr(idx), B, N, C, M, _ptr(out), _ptr(_err_word(dev)),
                                      _stream(dev))
        _lib.check(rc, "pn2_index_points")
        ctx.save_for_backward(idx)
        ctx.shape = (N, C)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        N, C = ctx.shape
        return index_points_backward(gout, idx, N, C), None


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,(K,)C]  (models/pointnet2_utils.py:43-60)."""
    dev = _dev(points, idx)
    out = _IndexPoints.apply(points.to(torch.float32).contiguous(), _i64c(idx))
    _after_fault_op(dev, "index_points")
    return out


class _GroupPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, points, idx, pad_to, inv_off=None, inv_ent=None, with_skip=False):
        dev = _dev(xyz, new_xyz, points, idx)
        lib = _lib.load()
        B, N, _ = xyz.shape
        _, S, K = idx.shape
        D = 0 if points is None else points.shape[2]
        ldg = _padded(3 + D, pad_to)
        out = torch.empty((B, S, K, ldg), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pn2_group_points(_ptr(xyz), _ptr(new_xyz), _ptr(points), _ptr(idx), B, N, S, K, D, _ptr(out), ldg,
                                      _ptr(_err_word(dev)), _stream(dev))
        _lib.check(rc, "pn2_group_points")
        ctx.has_inv = inv_off is not None
        ctx.save_for_backward(idx, *((inv_off, inv_ent) if ctx.has_inv else ()))
        ctx.shape = (N, D)
        ctx.with_skip = bool(with_skip)
        ctx.skip_owned = with_skip == "inplace"
        if with_skip:
            # second output: the points themselves, for the caller's OTHER use of them (the skip connection into
            # feature propagation) -- its gradient then arrives here and the scatter accumulates onto it
            return out, points.view_as(points)
        return out

    @staticmethod
    def backward(ctx, gout, gskip=None):
        idx = ctx.saved_tensors[0]
        inv = ctx.saved_tensors[1:3] if ctx.has_inv else None
        N, D = ctx.shape
        if D == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None, None, None, None
        if gout is None:
            return None, None, gskip, None, None, None, None, None
        return (None, None, index_points_backward(gout, idx, N, D, col0=3, inv=inv, into=gskip, into_owned=ctx.skip_owned),
                None, None, None, None, None)


def group_points(xyz, new_xyz, points, idx, pad_to=1, inv=None, with_skip=False):
    """[xyz[idx]-new_xyz, points[idx]] for a given idx (models/pointnet2_utils.py:127-132).
    inv = invert_index(idx, N): the backward then gathers instead of scatter-adding.
    with_skip: also returns `points` again (same storage); a caller that uses the points a second time (skip connection)
    through THAT tensor gets both gradients summed inside the scatter instead of by a zero fill + add.  The scatter
    accumulates onto a COPY of the skip gradient; with_skip="inplace" is the caller's statement that the gradient arriving
    for the second output has no other reader (a fresh tensor from the consumer's backward, as in the network's own wiring:
    mlp._MLPStack.backward hands over a tensor it just allocated) -- the scatter then adds onto it directly."""
    dev = _dev(xyz, new_xyz, points, idx)
    if points is not None:
        points = points.to(torch.float32).contiguous()
    io, ie = inv if inv is not None else (None, None)
    out = _GroupPoints.apply(_f32c(xyz), _f32c(new_xyz), points, _i64c(idx), pad_to, io, ie,
                             (with_skip if with_skip == "inplace" else bool(with_skip)) if points is not None else False)
    _after_fault_op(dev, "group_points")
    if with_skip and points is None:
        return out, None
    return out


# ---------------------------------------------------------------------------------- interpolation
# Graph captures of this package run inside capture_region(): Python's cyclic garbage collector is switched off for their
# duration.  A collection that starts in the middle of a capture finalises whatever garbage is around -- an earlier trainer's
# captured graphs, events, pool memory -- and HIP calls made from those finalisers are illegal while a stream is capturing:
# the process aborts (seen twice in the full GPU suite, "Fatal Python error: Aborted ... Garbage-collecting" inside a capture).
_CAPTURES_UNDERWAY = 0


class capture_region:
    def __enter__(self):
        import gc
        global _CAPTURES_UNDERWAY
        # (a nested region -- a capture inside a capture -- must not collect: it IS inside a capture)
        if _CAPTURES_UNDERWAY == 0 and os.environ.get("PN2_LAB_NO_COLLECT_BEFORE_CAPTURE", "0") != "1":   # (lab: A/B of it)
            gc.collect()                              # what is garbage now goes before the capture, not inside it
            if os.environ.get("PN2_LAB_EMPTY_CACHE_BEFORE_CAPTURE", "0") == "1" and torch.cuda.is_available():
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        self._was = gc.isenabled()
        gc.disable()
        _CAPTURES_UNDERWAY += 1
        return self

    def __exit__(self, *exc):
        import gc
        global _CAPTURES_UNDERWAY
        _CAPTURES_UNDERWAY -= 1
        if self._was and _CAPTURES_UNDERWAY == 0:
            gc.enable()
        return False


def capturing():
    return _CAPTURES_UNDERWAY > 0


# PN2_LAB_SKIP_NN=<N>: lab switch -- 3-NN tables with at least N queries are computed once and then reused (WRONG results for
# every later batch; it prices the launch for tools/ab_switch.sh, nothing else).  PN2_LAB_SKIP=inv,deep: the same for the
# transposed index tables and for the sampling + ball queries of levels 2-4 (models/pointnet2_utils.py).
_LAB_SKIP_NN = int(os.environ.get("PN2_LAB_SKIP_NN", "0"))
_LAB_NN_CACHE = {}
_LAB_SKIP = set(t for t in os.environ.get("PN2_LAB_SKIP", "").split(",") if t)
_LAB_CACHE = {}


def lab_cached(tag, key, fn):
    """fn() -- or, with `tag` in PN2_LAB_SKIP, what fn() returned the first time it was called with this key."""
    if tag not in _LAB_SKIP:
        return fn()
    k = (tag,) + tuple(key)
    if k not in _LAB_CACHE:
        _LAB_CACHE[k] = fn()
    return _LAB_CACHE[k]


def three_nn(xyz1, xyz2, want_dist=False):
    """xyz1 [B,N,3] queries, xyz2 [B,S,3] -> idx3 [B,N,3] int64, weight3 [B,N,3] (, dist3)."""
    dev = _dev(xyz1, xyz2)
    lib = _lib.load()
    xyz1, xyz2 = _f32c(xyz1), _f32c(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx3 = torch.empty((B, N, 3), dtype=torch.int64, device=dev)
    w3 = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    d3 = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if want_dist else None
    if _LAB_SKIP_NN and not want_dist and N >= _LAB_SKIP_NN:
        # measurement only (wrong for any batch but the first): what the table's launch costs the step beside it
        key = (B, N, S)
        if key in _LAB_NN_CACHE:
            return _LAB_NN_CACHE[key]
    with torch.cuda.device(dev):
        rc = lib.pn2_three_nn(_ptr(xyz1), _ptr(xyz2), B, N, S, _ptr(idx3), _ptr(d3), _ptr(w3), _stream(dev))
    if _LAB_SKIP_NN and not want_dist and N >= _LAB_SKIP_NN:
        _LAB_NN_CACHE[(B, N, S)] = (idx3, w3)
    _lib.check(rc, "pn2_three_nn")
    return (idx3, w3, d3) if want_dist else (idx3, w3)


def three_nn_many(pairs):
    """three_nn of several (xyz1 [B,N,3], xyz2 [B,S,3]) pairs of one batch size in ONE launch -> [(idx3, weight3), ...]."""
    import ctypes
    if not pairs:
        return []
    dev = _dev(*[t for p in pairs for t in p])
    lib = _lib.load()
    pairs = [(_f32c(a), _f32c(b)) for a, b in pairs]
    B = pairs[0][0].shape[0]
    n = len(pairs)
    if n > 8 or any(a.shape[0] != B or b.shape[0] != B for a, b in pairs):
        return [three_nn(a, b) for a, b in pairs]
    idx3 = [torch.empty((B, a.shape[1], 3), dtype=torch.int64, device=dev) for a, _ in pairs]
    w3 = [torch.empty((B, a.shape[1], 3), dtype=torch.float32, device=dev) for a, _ in pairs]
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    with torch.cuda.device(dev):
        rc = lib.pn2_three_nn_many(n, vp(*[a.data_ptr() for a, _ in pairs]), vp(*[b.data_ptr() for _, b in pairs]), B,
                                   ci(*[a.shape[1] for a, _ in pairs]), ci(*[b.shape[1] for _, b in pairs]),
                                   vp(*[t.data_ptr() for t in idx3]), vp(*[t.data_ptr() for t in w3]), _stream(dev))
    _lib.check(rc, "pn2_three_nn_many")
    return list(zip(idx3, w3))


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points2, idx3, weight3, inv_off=None, inv_ent=None):
        dev = _dev(points2, idx3, weight3)
        lib = _lib.load()
        B, S, D = points2.shape
        N = idx3.shape[1]
        out = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pn2_three_interpolate(_ptr(points2), _ptr(idx3), _ptr(weight3), B, N, S, D, _ptr(out),
                                           _stream(dev))
        _lib.check(rc, "pn2_three_interpolate")
        ctx.has_inv = inv_off is not None
        ctx.save_for_backward(idx3, weight3, *((inv_off, inv_ent) if ctx.has_inv else ()))
        ctx.shape = (B, N, S, D)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx3, weight3 = ctx.saved_tensors[:2]
        B, N, S, D = ctx.shape
        dev = gout.device
        lib = _lib.load()
        gout = _f32c(gout)
        if ctx.has_inv:
            return _gather_sum(gout, N, 0, ctx.saved_tensors[2:4], weight3, 3, S, D), None, None, None, None
        g2 = torch.zeros((B, S, D), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pn2_three_interpolate_backward(_ptr(gout), _ptr(idx3), _ptr(weight3), B, N, S, D, _ptr(g2),
                                                    _stream(dev))
        _lib.check(rc, "pn2_three_interpolate_backward")
        return g2, None, None, None, None


def three_interpolate(points2, idx3, weight3, inv=None):
    """sum_k points2[idx3[...,k]] * weight3[...,k]  (models/pointnet2_utils.py:303).
    inv = invert_index(idx3, S): the backward then gathers instead of scatter-adding."""
    io, ie = inv if inv is not None else (None, None)
    return _ThreeInterpolate.apply(points2.to(torch.float32).contiguous(), _i64c(idx3), _f32c(weight3), io, ie)


# ---------------------------------------------------------------------------------- loop glue
def input_blocks(x, channel_first=True, angles=None):
    """Input preparation of a training step on the device (localfunctions.py:205-209): x [B,C,N] (channel_first) or
    [B,N,C]; angles [B] radians (None: no rotation) rotate the xyz columns of each block about the up axis
    (provider.rotate_point_cloud_z).  -> (pts [B,N,C] channel-last rows, xyz [B,N,3])."""
    dev = _dev(x, angles)
    lib = _lib.load()
    x = _f32c(x)
    if channel_first:
        B, C, N = x.shape
    else:
        B, N, C = x.shape
    if angles is not None:
        angles = _f32c(angles)
        if angles.shape != (B,):
            raise ValueError("angles must be [B]")
    pts = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    xyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_input_blocks(_ptr(x), 1 if channel_first else 0, B, N, C, _ptr(angles), _ptr(pts), _ptr(xyz), _stream(dev))
    _lib.check(rc, "pn2_input_blocks")
    return pts, xyz


class SegMetrics:
    """Accuracy / IoU counters of the reference loops (localfunctions.py:214, 220-223, 271-289) kept on the device:
    add() enqueues one kernel per batch (no host synchronisation, hipGraph-capture safe), read() fetches the int64
    counters once -- per epoch -- and derives what the loops log."""

    def __init__(self, num_classes, device):
        self.C = int(num_classes)
        self.counters = torch.zeros(2 + 3 * self.C, dtype=torch.int64, device=device)

    def reset(self):
        self.counters.zero_()

    def add(self, logp, target):
        dev = _dev(logp, target, self.counters)
        lib = _lib.load()
        logp = _f32c(logp).reshape(-1, self.C)
        target = _i64c(target).reshape(-1)
        if target.numel() != logp.shape[0]:
            raise ValueError("logp rows and targets differ")
        with torch.cuda.device(dev):
            rc = lib.pn2_seg_metrics(_ptr(logp), _ptr(target), logp.shape[0], self.C, _ptr(self.counters), _stream(dev))
        _lib.check(rc, "pn2_seg_metrics")

    def read(self):
        c = self.counters.cpu().numpy().astype("float64")
        C = self.C
        seen_c, correct_c, union_c = c[2:2 + C], c[2 + C:2 + 2 * C], c[2 + 2 * C:2 + 3 * C]
        return {"correct": int(c[0]), "seen": int(c[1]), "accuracy": c[0] / max(c[1], 1.0),
                "class_seen": seen_c, "class_correct": correct_c, "class_union": union_c,
                "mIoU": float((correct_c / (union_c + 1e-6)).mean()),                      # localfunctions.py:283
                "avg_class_acc": float((correct_c / (seen_c + 1e-6)).mean())}              # :287-288
