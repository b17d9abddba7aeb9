"""Data-parallel training step for pointnet2_sem_seg: one process per GPU, replicas of the
968,914-parameter model, independent 4096-point blocks per rank, ONE all-reduce (RCCL over xGMI
when the backend is "nccl") of a flat fp32 gradient buffer per step.

The reference has no distributed code (single process, sem_seg_training.py:374); the step body
restates localfunctions.py:203-218: zero_grad -> forward -> nll_loss(weight) -> backward -> Adam
(sem_seg_training.py:576-582).  BatchNorm statistics stay rank-local (no SyncBN upstream)."""
import contextlib
import os

import torch
import torch.distributed as dist

from . import head, mlp

# LAB switch, measurement only (the step is then WRONG for any batch but the one it was captured on): the captured step
# keeps the first batch's geometry pyramid and has no geometry branch -- what the main branch costs when nothing runs beside it
_FREEZE_GEOMETRY = os.environ.get("PN2_LAB_FREEZE_GEOMETRY", "0") == "1"
# The next batch's pyramid is copied into the static buffers by the MAIN branch after the graph's single join (fork at the
# start, join after backward: -9 us per step); PN2_HANDOVER_ON_MAIN=0: by the side branch behind a second cross-queue
# dependency, the optimizer running beside the copy (the form of rounds 1-2)
_HANDOVER_ON_MAIN = os.environ.get("PN2_HANDOVER_ON_MAIN", "1") == "1"
# The geometry of the next batch is a hipGraph of its own, replayed on the side stream (PN2_SEPARATE_GEOMETRY_GRAPH=0: a branch
# of the step's graph, the form of rounds 1-2).  The hipGraph executor keeps ONE branch of a graph on the launch stream and moves the other
# to a queue of its own -- for the step's graph the branch that moved was the main one, and every step then paid a cross-queue
# signal at its start (fork) and at its end (join); rocprofv3 shows the single-branch graph running 135 kernels back to back
# (2.7 us of gaps in 2365 us).  With two graphs the main stream only ever waits on an event recorded one step earlier, and the
# first level's grouped rows (25 MB) are written where the next step reads them: 2.59 -> 2.53 ms per step.
_SEPARATE_GEOMETRY_GRAPH = os.environ.get("PN2_SEPARATE_GEOMETRY_GRAPH", "1") == "1"
# PN2_LATE_SIDE_ENQUEUE=0: the geometry graph is enqueued ahead of the step's graph behind a cross-stream wait (the form of
# rounds 3 / early 4) instead of behind a host wait for the previous step's graph (SemSegTrainer._enqueue_geometry)
_LATE_SIDE_ENQUEUE = os.environ.get("PN2_LATE_SIDE_ENQUEUE", "1") == "1"
# PN2_PACK_IN_PLACE=0: the pyramid's small tensors are concatenated into a temporary and copied (two launches per part) instead of
# concatenated straight into the pyramid buffer
_PACK_IN_PLACE = os.environ.get("PN2_PACK_IN_PLACE", "1") == "1"
# lab (wrong results): the geometry graph is captured but never replayed -- the step graphs in their shipped form, alone
_LAB_NO_SIDE_REPLAY = os.environ.get("PN2_LAB_NO_SIDE_REPLAY", "0") == "1"
_LAB_SIDE_DELAY = int(os.environ.get("PN2_LAB_SIDE_DELAY_US", "0"))         # lab: the host enqueues the geometry graph this much later
# PN2_ALTERNATE_STEP_GRAPHS (default 1; single process, with the geometry graph): two captured step graphs that read the pyramid
# from two buffers in turn, so that the 27 MB hand-over copy (12 us) disappears from the main stream: 2.493 -> 2.480 ms per step,
# four alternating runs each (profiles/r04/ab_alternate_step_graphs.log).  0: one step graph and the copy.
_ALTERNATE_STEP_GRAPHS = os.environ.get("PN2_ALTERNATE_STEP_GRAPHS", "1") == "1"
_DEFER_DW = os.environ.get("PN2_DEFER_DW", "1") != "0"    # A/B switch: 0 = every stack sums its bottom layer's slabs at once


def rotate_z_(blocks_cf, angles=None):
    """On-device form of provider.rotate_point_cloud_z applied to the xyz channels of a batch
    (reference localfunctions.py:206, provider.py:66-84): per block a rotation about the up axis by
    `angles` [B] (default: uniform in [0, 2*pi), drawn on the device), x' = x*cos - y*sin,
    y' = x*sin + y*cos.  blocks_cf is [B,C,N] channel-first; rotated in place, no host round trip."""
    B = blocks_cf.shape[0]
    if angles is None:
        angles = torch.rand(B, device=blocks_cf.device) * (2.0 * 3.141592653589793)
    c, s = torch.cos(angles).view(B, 1), torch.sin(angles).view(B, 1)
    x, y = blocks_cf[:, 0, :].clone(), blocks_cf[:, 1, :].clone()
    blocks_cf[:, 0, :] = x * c - y * s                  # row-vector @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    blocks_cf[:, 1, :] = x * s + y * c
    return blocks_cf


def pack_segments(tensors, pads, out=None):
    """The tensors (None = absent) with their zero pads (uint8, None = none) as one uint8 buffer.  Concatenated as
    32-bit words when every segment allows it: torch's byte-wise cat moves one byte per thread (110 us for the 27 MB of
    a pyramid with the first level's grouped rows; 4x fewer elements this way).  out (uint8, exactly the packed size):
    the concatenation is written there by the cat kernel itself -- no packed temporary and no copy of it."""
    parts = []
    for t, pad in zip(tensors, pads):
        if t is not None:
            parts.append(t.contiguous().view(-1).view(torch.uint8))
        if pad is not None:
            parts.append(pad)
    words = all(p.numel() % 4 == 0 and p.data_ptr() % 4 == 0 for p in parts)
    if out is not None and _PACK_IN_PLACE and out.numel() == sum(p.numel() for p in parts) and out.is_contiguous():
        if words and out.data_ptr() % 4 == 0:
            torch.cat([p.view(torch.int32) for p in parts], out=out.view(torch.int32))
        else:
            torch.cat(parts, out=out)
        return out
    packed = torch.cat([p.view(torch.int32) for p in parts]).view(torch.uint8) if words else torch.cat(parts)
    if out is not None:
        out.copy_(packed)
        return out
    return packed


class FlatGradients:
    """Data-parallel gradient exchange as ONE collective: after backward the per-parameter
    gradients are packed into a contiguous fp32 buffer (one concat kernel, 3.88 MB for
    pointnet2_sem_seg), all-reduced, and the parameters' .grad become views of that buffer.
    With a single process nothing is packed or copied at all."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        self.numel = sum(p.numel() for p in self.params)
        self.buffer = None

    def zero(self):
        # None, not zeros: autograd then assigns each gradient instead of launching an add per parameter
        for p in self.params:
            p.grad = None
        self.buffer = None

    def pack(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        self.buffer = torch.cat([g.reshape(-1) for g in grads])
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.buffer[off:off + n].view_as(p)
            off += n
        return self.buffer

    def all_reduce_mean(self, group=None):
        """Sum over ranks, then divide by world size.  No-op for a single process."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1 and os.environ.get("PN2_FORCE_DP_PATH", "0") != "1":
            return
        flat = self.pack()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-08, weight_decay) of the reference loop
    (sem_seg_training.py:576-582) as ONE elementwise pass (pn2_adam_step): the parameters become views of one
    flat buffer (same order as FlatGradients packs their gradients), the moments, the step counter and the
    learning rate live on the device (hipGraph replay).  HIP device only."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        from . import _lib, mlp
        self._lib = _lib
        self.params = list(params)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                view = self.flat[off:off + k].view_as(p)
                view.copy_(p)
                p.data = view
                off += k
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.state = torch.zeros(3, dtype=torch.float32, device=dev)      # step, 1-beta1^t, sqrt(1-beta2^t)
        self.lr = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self._offsets = None

    def set_lr(self, lr):
        """New learning rate (the reference re-sets param_group['lr'] every epoch, localfunctions.py:187-190);
        a device scalar, so captured graphs pick it up."""
        self.lr.fill_(float(lr))

    def state_dict(self):
        """torch.optim.Adam's layout (what the reference stores as 'optimizer_state_dict', localfunctions.py:229-239,
        315-321): per parameter index {'step', 'exp_avg', 'exp_avg_sq'} -- copies, in the parameters' own shapes -- and one
        param_group with the hyper-parameters.  A checkpoint written here loads into torch.optim.Adam and back."""
        step = float(self.state[0].item())
        state, off = {}, 0
        for i, p in enumerate(self.params):
            k = p.numel()
            if step > 0:
                state[i] = {"step": torch.tensor(step), "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": float(self.lr.item()), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """The inverse: moments and step count from a torch.optim.Adam state dict over the same parameters in the same
        order (a parameter without state -- never stepped -- gets zeros).  In place: captured graphs keep reading the
        same buffers."""
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError("optimizer state for %d parameters, this optimizer has %d" % (len(ids), len(self.params)))
        steps, off = [], 0
        with torch.no_grad():
            for pid, p in zip(ids, self.params):
                k = p.numel()
                st = sd["state"].get(pid)
                if st is None:
                    self.exp_avg[off:off + k].zero_()
                    self.exp_avg_sq[off:off + k].zero_()
                else:
                    if st["exp_avg"].numel() != k:
                        raise ValueError("optimizer state %d has %d elements, the parameter %d" % (pid, st["exp_avg"].numel(), k))
                    self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.append(float(st["step"]))
                off += k
            if steps and min(steps) != max(steps):
                raise ValueError("parameters were stepped %g .. %g times: one flat pass has one step count" % (min(steps), max(steps)))
            self.state.zero_()
            self.state[0] = steps[0] if steps else 0.0       # the kernel derives the bias corrections from the count
            self.lr.fill_(float(groups[0]["lr"]))
        self.betas, self.eps, self.weight_decay = tuple(groups[0]["betas"]), groups[0]["eps"], groups[0]["weight_decay"]

    def step_scattered(self, grads, grad_scale=1.0):
        """The update with the gradients where backward left them (one tensor or None per parameter, in the order of
        `params`): no packing pass.  False when the library does not take this many tensors (pack and use step())."""
        import ctypes
        lib = self._lib.load()
        dev = self.flat.device
        n = len(self.params)
        if self._offsets is None:
            off = [0]
            for p in self.params:
                off.append(off[-1] + p.numel())
            self._offsets = (ctypes.c_longlong * (n + 1))(*off)
        held = [None if g is None else (g if (g.is_contiguous() and g.dtype == torch.float32) else g.float().contiguous())
                for g in grads]
        ptrs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in held])
        with torch.cuda.device(dev):
            rc = lib.pn2_adam_step_scattered(self.flat.data_ptr(), n, ptrs, self._offsets, self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.lr.data_ptr(), self.state.data_ptr(),
                                             self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale,
                                             torch.cuda.current_stream(dev).cuda_stream)
        if rc == self._lib.ERR_UNSUPPORTED:
            return False
        self._lib.check(rc, "pn2_adam_step_scattered")
        mlp.invalidate_eval_coefficients()                 # gamma / beta changed through raw pointers
        return True

    def step(self, flat_grad, grad_scale=1.0):
        lib = self._lib.load()
        dev = self.flat.device
        with torch.cuda.device(dev):
            rc = lib.pn2_adam_step(self.flat.data_ptr(), flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), self.flat.numel(), self.lr.data_ptr(), self.state.data_ptr(),
                                   self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale,
                                   torch.cuda.current_stream(dev).cuda_stream)
        self._lib.check(rc, "pn2_adam_step")
        mlp.invalidate_eval_coefficients()


class SemSegTrainer:
    """graphs=True (HIP device only): the step is captured into hipGraphs after a few eager
    warm-up steps and replayed -- one graph for zero_grad+forward+loss+backward(+gradient
    packing), the all-reduce issued eagerly between (world > 1), one graph for Adam; with a single
    process everything is one graph.  ~250 kernel launches per step otherwise keep the host as
    busy as the GPU."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-4, class_weight=None, group=None, graphs=False,
                 graph_warmup=3, prefetch_geometry=False, augment=False, metrics=False):
        """augment: rotate every block about the up axis by a fresh random angle per step, on the device, inside the
        input-preparation kernel (the reference does it on the host between two copies, localfunctions.py:205-208).
        metrics: accumulate the loop's accuracy / IoU counters on the device (self.metrics.read() once per epoch
        replaces the per-step .cpu() of localfunctions.py:214-223)."""
        self.model = model
        self.group = group
        self.grads = FlatGradients(model)
        self.class_weight = class_weight
        from .models.pointnet2_sem_seg import get_loss   # criterion of the reference loop (localfunctions.py:212)
        self.criterion = get_loss()
        on_gpu = next(model.parameters()).is_cuda
        self.graphs = bool(graphs) and on_gpu
        # HIP device: one flat Adam pass over the packed gradients; CPU (tests of the exchange): torch's Adam
        self.flat_adam = FlatAdam(self.grads.params, lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                  weight_decay=weight_decay) if on_gpu else None
        self.optimizer = None if on_gpu else torch.optim.Adam(self.grads.params, lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                                              weight_decay=weight_decay)
        # prefetch_geometry: the FPS / ball-query / 3-NN pyramid of the NEXT batch (a function of
        # its coordinates only) is computed on a side stream while this batch's MLP work runs:
        # FPS is a latency-bound chain that occupies 16 of 256 CUs.
        self.prefetch = bool(prefetch_geometry) and on_gpu and hasattr(model, "compute_geometry")
        self.augment = bool(augment) and on_gpu and hasattr(model, "prepare_input")
        self.metrics = None
        if metrics and on_gpu:
            from .ops import SegMetrics
            ncls = model.conv2.out_channels if hasattr(model, "conv2") else int(metrics)
            self.metrics = SegMetrics(ncls, next(model.parameters()).device)
        # PN2_SIDE_STREAM_PRIORITY: HIP priority of the geometry stream (0 = default = the main stream's; larger = lower)
        self._side = torch.cuda.Stream(priority=int(os.environ.get("PN2_SIDE_STREAM_PRIORITY", "0"))) if self.prefetch else None
        self._prepares = hasattr(model, "prepare_input")        # the pyramid hands the prepared input rows over too
        self._unit = torch.ones((), dtype=torch.float32, device=next(model.parameters()).device) if on_gpu else None
        self._geo_next = None            # pyramid computed for the coming step
        self._geo_next_src = None        # the batch it was computed from: (the tensor itself, its version)
        self._captured_mode = None       # (exchange, world) the graphs were captured for
        self._geo_event = None
        self._geo_cur = None             # (graph mode) static buffers the forward reads
        self._static_next_x = None
        self._graph_warmup = graph_warmup
        self._eager_steps = 0
        self._g_fwd_bwd = self._g_opt = self._g_geo = self._alt = None
        self._static_x = self._static_y = self._static_loss = None
        self._tap = None                 # test hook: called with clones of the pyramid tensors a replay is about to read

    def broadcast_parameters(self, src=0):
        """Replicas start identical: rank `src`'s parameters and buffers go to every rank."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)

    def __del__(self):
        # a captured graph must not be destroyed while a replay of it is still running: the geometry graph of the last step
        # runs on the side stream for about a millisecond after step() has returned -- wait for the device before the graphs go
        try:
            from . import ops
            if self._g_fwd_bwd is not None and not ops.capturing():       # (a device wait is illegal while a stream captures)
                torch.cuda.synchronize()
        except Exception:
            pass

    def prepare(self, blocks_cf, target):
        """Capture the hipGraphs NOW, without changing the model: `graph_warmup` dry forward/backward
        passes (allocator and lazy initialisations settle; BatchNorm buffers are restored afterwards, no
        optimizer step), then the capture.

        Why: graphs captured while an RCCL communicator exists replay 9 % slower on this stack (measured on
        MI355X, ROCm 7.2: 3.43 -> 3.73 ms; the communicator's streams change how the graph's kernels are
        spread over the hardware queues), while a communicator created AFTER the capture costs nothing.  A
        data-parallel job therefore calls prepare() before its first collective (and broadcast_parameters()
        after it); every rank runs the same dry passes, so replicas stay identical."""
        if not self.graphs or self._g_fwd_bwd is not None:
            return
        self.model.train()
        saved = [b.detach().clone() for b in self.model.buffers()]
        for _ in range(max(1, self._graph_warmup)):
            self._forward_backward(blocks_cf, target)
        with torch.no_grad():
            for b, s in zip(self.model.buffers(), saved):
                b.copy_(s)
        self.grads.zero()
        torch.cuda.synchronize()
        self._geo_next = None
        self._eager_steps = self._graph_warmup
        self._capture(blocks_cf, target)

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def set_lr(self, lr):
        """Per-epoch learning rate of the reference loop (localfunctions.py:187-190); graph-safe (device scalar)."""
        if self.flat_adam is not None:
            self.flat_adam.set_lr(lr)
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = lr

    def set_bn_momentum(self, momentum):
        """Per-epoch BatchNorm momentum of the reference loop (localfunctions.py:191-195); graph-safe: the captured
        bn_finalize launches read it from device words (mlp.momentum_word)."""
        mlp.set_bn_momentum(self.model, momentum)

    @staticmethod
    def _identity(t):
        """What "the batch the pyramid was computed from" means: THE tensor object (held, so that the allocator cannot hand
        its address to another batch while the pyramid is pending -- sampler and .to() outputs are written by raw kernels
        and all carry version 0) at the version it had."""
        return None if t is None else (t, t._version)

    @staticmethod
    def _same_batch(src, t):
        if src is None or t is None or src[1] != t._version:
            return False
        held = src[0]                    # alive, so its storage is: a tensor with the same address aliases the same memory
        return held is t or (held.data_ptr() == t.data_ptr() and held.shape == t.shape and held.stride() == t.stride()
                             and held.dtype == t.dtype)

    def drop_prefetched(self):
        """Forget the pyramid computed for an announced batch that will not come (end of an epoch)."""
        self._geo_next = None
        self._geo_next_src = None

    def _exchange(self):
        """True when gradients go through the packed-buffer all-reduce.  PN2_FORCE_DP_PATH=1 takes
        that path with a single rank too (rehearsal of the N > 1 code on a one-GPU box)."""
        return self._world() > 1 or (os.environ.get("PN2_FORCE_DP_PATH", "0") == "1"
                                     and dist.is_available() and dist.is_initialized())

    def _draw_angles(self, blocks_cf):
        return torch.rand(blocks_cf.shape[0], device=blocks_cf.device) * 6.283185307179586     # provider.py:76

    def _geometry_of(self, blocks_cf):
        """The pyramid of a batch; with augmentation the rotated input it was computed on travels with it (last two
        entries: rows [B,N,C], coordinates [B,N,3]) so that the forward one step later runs on the same rotation."""
        with torch.no_grad():
            if not self._prepares:
                return self.model.compute_geometry(blocks_cf)
            # the input rows are laid out (and rotated) on this branch too, and the first level's grouped rows come from
            # the launch that finds its indices: the critical branch starts at the first GEMM
            prepared = self.model.prepare_input(blocks_cf, self._draw_angles(blocks_cf) if self.augment else None)
            return self.model.compute_geometry(prepared=prepared, group_first=True) + [prepared[0], prepared[1]]

    def _launch_prefetch(self, next_blocks_cf):
        """Enqueue the geometry pyramid of `next_blocks_cf` on the side stream."""
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            geo = self._geometry_of(next_blocks_cf)
        return geo

    def _fill_next_pyramid(self, flat=None):
        """The pyramid of `_static_next_x` into a static buffer (default: the second one).  The first level's grouped rows (25 of the 27 MB)
        are written by the query launch itself (ops.place_next_grouped); the small tensors before and behind them are
        concatenated and copied -- packing everything and copying the pack moved the rows twice."""
        from . import ops
        flat = self._geo_next_flat if flat is None else flat
        big = max(range(len(self._geo_shapes)), key=lambda i: 0 if self._geo_shapes[i] is None else self._geo_offs[i + 1] - self._geo_offs[i])
        dtype, shape = self._geo_shapes[big]
        o0, o1 = self._geo_offs[big], self._geo_offs[big + 1]
        pad = 0 if self._geo_pads[big] is None else self._geo_pads[big].numel()
        view = flat[o0:o1 - pad].view(dtype).view(shape)
        placeable = dtype == torch.float32 and len(shape) == 4
        if placeable:
            ops.place_next_grouped(view)
        try:
            geo = self._geometry_of(self._static_next_x)
        finally:
            ops.place_next_grouped(None)                    # an offer nobody took must not reach an unrelated call
        if placeable and geo[big] is not None and geo[big].data_ptr() == view.data_ptr():
            if big > 0:
                pack_segments(geo[:big], self._geo_pads[:big], out=flat[:o0])
            if big + 1 < len(geo):
                pack_segments(geo[big + 1:], self._geo_pads[big + 1:], out=flat[o1:])
            if pad:
                flat[o1 - pad:o1].zero_()
        else:
            flat.copy_(self._pack_geometry(geo))

    def _views_of(self, flat):
        """The pyramid's tensors as views of a flat byte buffer with the layout fixed at capture time."""
        views = []
        for i, sh in enumerate(self._geo_shapes):
            if sh is None:
                views.append(None)
                continue
            pad = 0 if self._geo_pads[i] is None else self._geo_pads[i].numel()
            views.append(flat[self._geo_offs[i]:self._geo_offs[i + 1] - pad].view(sh[0]).view(sh[1]))
        return views

    def _pack_geometry(self, geo):
        """The pyramid's tensors as one uint8 buffer (segment layout fixed at capture time)."""
        return pack_segments(geo, self._geo_pads)

    def _forward_backward(self, blocks_cf, target, geometry=None):
        self.grads.zero()
        if geometry is not None and self._prepares:
            prepared, geometry = (geometry[-2], geometry[-1]), geometry[:-2]
            pred, _ = self.model(blocks_cf, geometry=geometry, prepared=prepared)
        elif self.augment:
            prepared = self.model.prepare_input(blocks_cf, self._draw_angles(blocks_cf))
            pred, _ = self.model(blocks_cf, geometry=geometry, prepared=prepared)
        else:
            pred, _ = self.model(blocks_cf) if geometry is None else self.model(blocks_cf, geometry=geometry)
        loss = self.criterion(pred.reshape(-1, pred.shape[-1]), target.reshape(-1), None, self.class_weight)
        # the seed of backward is a resident 1.0 (loss.backward() alone fills a fresh ones_like every step: one launch)
        # the weight gradients of every stack's bottom layer are summed in ONE launch at the end of backward (nobody reads
        # them before the optimizer; every .grad is None here, so autograd keeps the tensors it is handed)
        with (mlp.deferred_weight_sums(self.grads.params) if _DEFER_DW else contextlib.nullcontext()):
            loss.backward(self._unit if (self._unit is not None and loss.dim() == 0 and loss.dtype == self._unit.dtype) else None)
        if self.metrics is not None:
            self.metrics.add(pred.detach(), target)      # one kernel, no host sync (localfunctions.py:220-223)
        return loss.detach()

    def _optimizer_step(self, grad_scale=1.0):
        if self.flat_adam is not None:
            # no exchange packed the gradients: the update reads them where backward left them
            if self.grads.buffer is None and not self.flat_adam.step_scattered([p.grad for p in self.grads.params], grad_scale):
                self.grads.pack()
            if self.grads.buffer is not None:
                self.flat_adam.step(self.grads.buffer, grad_scale)
            self.grads.buffer = None
        else:
            self.optimizer.step()

    def _eager_step(self, blocks_cf, target, next_blocks_cf=None):
        geo = None
        if self.prefetch:
            main = torch.cuda.current_stream()
            # the prefetched pyramid belongs to the tensor it was computed from: any other batch (or the same tensor
            # modified in place since) gets its own pyramid now instead of silently grouping with foreign indices
            if self._geo_next is None or not self._same_batch(self._geo_next_src, blocks_cf):
                self._geo_next = self._launch_prefetch(blocks_cf)
            main.wait_stream(self._side)
            geo, self._geo_next = self._geo_next, None
            nxt = blocks_cf if next_blocks_cf is None else next_blocks_cf
            self._geo_next = self._launch_prefetch(nxt)  # overlaps with everything below
            self._geo_next_src = self._identity(nxt)
            for t in self._geo_next:
                if t is not None:
                    t.record_stream(self._side)
        loss = self._forward_backward(blocks_cf, target, geo)
        self.grads.all_reduce_mean(self.group)          # packs (grads.buffer) when there is an exchange
        self._optimizer_step()
        return loss

    def _capture(self, blocks_cf, target):
        from . import ops
        with ops.capture_region():                      # no garbage collection inside a capture (ops.capture_region)
            self._capture_graphs(blocks_cf, target)

    def _capture_graphs(self, blocks_cf, target):
        exchange = self._exchange()
        mlp.ensure_momentum_words(self.model)                   # outside the graph: replays then follow set_bn_momentum()
        head.ensure_ticket_words(blocks_cf.device)
        self._captured_mode = (exchange, self._world())
        self._geo_next_src = self._identity(blocks_cf)          # (prefetch) `cur` will hold this batch's pyramid
        self._static_x = blocks_cf.clone()
        self._static_y = target.clone()
        if self.prefetch:
            # static pyramid buffer `cur`: read by forward and backward of a replay; the side branch of the
            # same replay computes the next batch's pyramid, and once backward is done it is copied into `cur`
            # (one 6 us copy: see _HANDOVER_ON_MAIN)
            self._static_next_x = blocks_cf.clone()
            torch.cuda.current_stream().wait_stream(self._side)
            first = self._geometry_of(self._static_x)
            # all pyramid tensors live in ONE byte buffer (16-byte aligned segments), so the hand-over after
            # backward is a single copy instead of one small copy kernel per tensor (24 x 4 us on the step's tail)
            self._geo_pads, off = [], 0
            for t in first:
                nbytes = 0 if t is None else t.numel() * t.element_size()
                pad = (-nbytes) % 16
                self._geo_pads.append(torch.zeros(pad, dtype=torch.uint8, device=blocks_cf.device) if pad else None)
                off += nbytes + pad
            self._geo_flat = torch.empty(off, dtype=torch.uint8, device=blocks_cf.device)
            self._geo_flat.copy_(self._pack_geometry(first))
            self._geo_cur, self._geo_offs, off = [], [], 0
            for t, padt in zip(first, self._geo_pads):
                self._geo_offs.append(off)
                if t is None:
                    self._geo_cur.append(None)
                    continue
                nbytes = t.numel() * t.element_size()
                self._geo_cur.append(self._geo_flat[off:off + nbytes].view(t.dtype).view(t.shape))
                off += nbytes + (0 if padt is None else padt.numel())
            self._geo_offs.append(off)
            self._geo_shapes = [None if t is None else (t.dtype, tuple(t.shape)) for t in first]
            torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        separate = self.prefetch and _SEPARATE_GEOMETRY_GRAPH and not _FREEZE_GEOMETRY
        self._g_geo = None
        if separate:
            # the next batch's pyramid as a graph of its OWN on the side stream, written into a second static buffer; the step's
            # graph then has ONE branch and stays on the stream it is launched on (see _SEPARATE_GEOMETRY_GRAPH)
            self._geo_next_flat = self._geo_flat.clone()
            self._geo_ready = torch.cuda.Event()
            self._inputs_ready = torch.cuda.Event()
            self._step_ends, self._step_end, self._step_parity = [torch.cuda.Event(), torch.cuda.Event()], None, 0
            self._side.wait_stream(torch.cuda.current_stream())
            self._g_geo = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_geo, stream=self._side):     # a pool of its own: the two graphs run side by side
                self._fill_next_pyramid()
            torch.cuda.current_stream().wait_stream(self._side)
            torch.cuda.synchronize()
        # Two alternating step graphs (single process only): step graph p reads pyramid buffer p while geometry graph p fills
        # buffer 1 - p for the step after it -- no 27 MB hand-over copy on the main stream (see _ALTERNATE_STEP_GRAPHS)
        self._alt = None
        alternate = separate and _ALTERNATE_STEP_GRAPHS and not exchange
        if alternate:
            bufs = [self._geo_flat, self._geo_next_flat]
            views = [self._geo_cur, self._views_of(self._geo_next_flat)]
            geo_pool = self._g_geo.pool()
            self._side.wait_stream(torch.cuda.current_stream())
            g_geo1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_geo1, stream=self._side, pool=geo_pool):
                self._fill_next_pyramid(bufs[0])
            torch.cuda.current_stream().wait_stream(self._side)
            torch.cuda.synchronize()
            self._alt = {"geo": [self._g_geo, g_geo1], "main": [None, None], "loss": [None, None], "bufs": bufs, "views": views, "p": 0}
        self._g_fwd_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_fwd_bwd, pool=pool):
            geo = new_geo = None
            if (self.prefetch and _FREEZE_GEOMETRY) or separate:
                geo = self._geo_cur                                       # no side branch in this graph
            elif self.prefetch:
                new_geo = self._launch_prefetch(self._static_next_x)      # fork: side branch of the graph
                with torch.cuda.stream(self._side):
                    new_flat = self._pack_geometry(new_geo)
                geo = self._geo_cur
            self._static_loss = self._forward_backward(self._static_x, self._static_y, geo)
            if separate:
                pass
            elif self.prefetch and not _FREEZE_GEOMETRY and _HANDOVER_ON_MAIN:
                torch.cuda.current_stream().wait_stream(self._side)       # the only join; backward no longer reads `cur`
                self._geo_flat.copy_(new_flat)
            elif self.prefetch and not _FREEZE_GEOMETRY:
                self._side.wait_stream(torch.cuda.current_stream())       # backward no longer reads `cur`
                with torch.cuda.stream(self._side):
                    self._geo_flat.copy_(new_flat)
            flat = None
            if exchange or not self.flat_adam.step_scattered([p.grad for p in self.grads.params]):
                flat = self.grads.pack()                # .grad become views of one flat buffer
                if not exchange:
                    self.flat_adam.step(flat)
            if self.prefetch and not _FREEZE_GEOMETRY and not _HANDOVER_ON_MAIN and not separate:
                torch.cuda.current_stream().wait_stream(self._side)       # join
        if alternate:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, pool=pool):
                loss1 = self._forward_backward(self._static_x, self._static_y, views[1])
                if not self.flat_adam.step_scattered([p.grad for p in self.grads.params]):
                    self.flat_adam.step(self.grads.pack())
            self._alt["main"] = [self._g_fwd_bwd, g1]
            self._alt["loss"] = [self._static_loss, loss1]
        if exchange:
            # the all-reduce (sum) runs eagerly between the two graphs; the 1/world goes into the Adam pass
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, pool=pool):
                self.flat_adam.step(flat, 1.0 / self._world())

    def _enqueue_geometry(self, graph, nxt):
        """The side stream's work of this step (the next batch's pyramid), enqueued BEHIND the step's graph.
        The geometry graph may not start before the PREVIOUS step's graph has finished (that graph reads the pyramid buffer
        this one fills).  Enqueued ahead of time behind a cross-stream wait, that dependency sits at the head of the side
        queue as an unsatisfied barrier for as long as the host runs ahead of the GPU -- and a step graph runs 95 us longer
        beside a queue that is blocked like that (tools/step_events.py: 2.395 against 2.300 ms with nothing at all running on
        the side stream).  So the host waits here until the previous step's graph has finished, then enqueues: nothing is
        ever pending on the side queue.  The GPU does not starve -- this step's graph is already queued -- and the host stays
        at most one step ahead; a caller who synchronises every step never waits here."""
        main = torch.cuda.current_stream()
        if _LATE_SIDE_ENQUEUE:
            prev, self._step_end = self._step_end, self._step_ends[self._step_parity]
            self._step_parity ^= 1
            self._step_end.record(main)                         # this step's graph is in the queue up to here
            if prev is not None:
                prev.synchronize()                              # host: the previous step's graph has finished
                if _LAB_SIDE_DELAY:                             # lab: the geometry graph starts this many us into the step
                    import time
                    t_end = time.perf_counter() + _LAB_SIDE_DELAY * 1e-6
                    while time.perf_counter() < t_end:
                        pass
            self._side.wait_event(self._inputs_ready)           # (fires at once: it sits in front of this step's graph)
            with torch.cuda.stream(self._side):
                self._static_next_x.copy_(nxt)
            if nxt.is_cuda:
                nxt.record_stream(self._side)                   # the caller may free it while the side stream still reads it
        with torch.cuda.stream(self._side):
            if not _LAB_NO_SIDE_REPLAY:
                graph.replay()
            self._geo_ready.record()

    def step(self, blocks_cf, target, next_blocks_cf=None):
        """blocks_cf [B,C,N] (channel-first like the reference loop, localfunctions.py:209),
        target [B,N] int64; next_blocks_cf (optional, with prefetch_geometry) is the batch the
        NEXT call will train on (default: the same batch again).  Returns the (rank-local) loss
        tensor.  No host synchronisation with THIS step; with the geometry graph the call returns once the PREVIOUS
        step's graph has finished (_enqueue_geometry: the host stays at most one step ahead of the GPU)."""
        self.model.train()
        mlp.invalidate_eval_coefficients()                 # a replayed graph rewrites weights and running statistics
        if not self.graphs:
            return self._eager_step(blocks_cf, target, next_blocks_cf)
        if self._g_fwd_bwd is None:
            if self._eager_steps < self._graph_warmup:  # allocator / MIOpen / lazy inits settle first
                self._eager_steps += 1
                return self._eager_step(blocks_cf, target, next_blocks_cf)
            torch.cuda.synchronize()
            self._geo_next = None
            self._capture(blocks_cf, target)
        if self._captured_mode != (self._exchange(), self._world()):
            raise RuntimeError("the step was captured for %s (gradient exchange, world size) but now runs with %s: "
                               "capture (prepare()) before init_process_group creates no all-reduce -- call prepare() "
                               "after the process group exists, or build a new SemSegTrainer"
                               % (self._captured_mode, (self._exchange(), self._world())))
        # With a prefetched pyramid that carries the prepared input rows the replay never reads `_static_x` (the pyramid of
        # an announced batch was computed from `_static_next_x` one step ago): its copy is left out, 2.4 MB per step
        announced = self.prefetch and self._prepares and self._same_batch(self._geo_next_src, blocks_cf)
        if blocks_cf.data_ptr() != self._static_x.data_ptr() and not announced:
            self._static_x.copy_(blocks_cf)
        if target.data_ptr() != self._static_y.data_ptr():
            self._static_y.copy_(target)
        if self.prefetch:
            main = torch.cuda.current_stream()
            alt = self._alt
            if self._g_geo is not None:
                # the side stream's graph of the previous step reads `_static_next_x` and fills the other pyramid buffer: both
                # are touched below.  Its event was recorded one step ago and has long fired -- the main stream does not stall
                main.wait_event(self._geo_ready)
            if not self._same_batch(self._geo_next_src, blocks_cf):
                # the pyramid in `cur` was computed for another batch (the caller did not announce this one as
                # next_blocks_cf): compute this batch's pyramid now, on the main stream, before the replay reads it
                (self._geo_flat if alt is None else alt["bufs"][alt["p"]]).copy_(self._pack_geometry(self._geometry_of(self._static_x)))
            elif self._g_geo is not None and alt is None:
                self._geo_flat.copy_(self._geo_next_flat)    # the pyramid the side graph left for this batch
            nxt = blocks_cf if next_blocks_cf is None else next_blocks_cf
            self._geo_next_src = self._identity(nxt)
            if self._g_geo is not None and not _LATE_SIDE_ENQUEUE:
                # the next input is copied on the SIDE stream, in front of the graph that reads it (the previous replay of that
                # graph, the buffer's only other reader, is ahead of it on the same stream): 2.4 MB less on the main stream
                self._side.wait_stream(main)                    # `nxt` is ready, the second pyramid buffer is free
                with torch.cuda.stream(self._side):
                    self._static_next_x.copy_(nxt)
                if nxt.is_cuda:
                    nxt.record_stream(self._side)               # the caller may free it while the side stream still reads it
            elif self._g_geo is not None:
                self._inputs_ready.record(main)                 # `nxt` and the fix-ups above are in the main queue up to here
            else:
                self._static_next_x.copy_(nxt)
        if self.prefetch and self._tap is not None:
            read = self._geo_cur if self._alt is None else self._alt["views"][self._alt["p"]]
            self._tap([None if t is None else t.clone() for t in read])     # enqueued behind the fix-ups, in front of the replay
        if self.prefetch and self._alt is not None:
            alt = self._alt
            p = alt["p"]
            alt["main"][p].replay()                             # reads pyramid buffer p
            self._enqueue_geometry(alt["geo"][p], nxt)          # fills buffer 1 - p for the next step
            self._static_loss = alt["loss"][p]
            alt["p"] = p ^ 1
        else:
            self._g_fwd_bwd.replay()
            if self.prefetch and self._g_geo is not None:
                self._enqueue_geometry(self._g_geo, nxt)
        if self._g_opt is not None:
            dist.all_reduce(self.grads.buffer, op=dist.ReduceOp.SUM, group=self.group)
            self._g_opt.replay()
        return self._static_loss


def epoch_schedule(epoch, learning_rate=1e-3, lr_decay=0.7, step_size=10):
    """(lr, BatchNorm momentum) of an epoch, localfunctions.py:168-193: lr = max(lr0 * decay^(epoch // step), 1e-5);
    momentum = max(0.1 * 0.5^(epoch // step), 0.01)."""
    lr = max(learning_rate * (lr_decay ** (epoch // step_size)), 1e-5)
    momentum = max(0.1 * (0.5 ** (epoch // step_size)), 0.01)
    return lr, momentum


def _mix(*words):
    """64-bit counter hash (splitmix64 finaliser over the words): the seeds of the loops below."""
    x = 0x9E3779B97F4A7C15
    for w in words:
        x = (x ^ (int(w) & 0xFFFFFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9 & 0xFFFFFFFFFFFFFFFF
        x = (x ^ (x >> 31)) * 0x94D049BB133111EB & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 29
    return x


def batch_plan(room_sizes, batch_size, seed, epoch, step, rank=0):
    """Which room every block of a batch comes from, and the sampler seed per room -- a pure function of
    (seed, epoch, step, rank).  The reference replicates each room's index in proportion to its point count and shuffles
    the list (room_idxs, sem_seg_training.py:184-193; DataLoader shuffle=True): a block's room is drawn with probability
    num_points_room / total, and a batch mixes rooms.  The rank enters the hash, so the replicas of a data-parallel job
    draw DIFFERENT blocks (rank r takes its own share of every global batch).
    -> (counts per room [R], seeds per room [R])."""
    import numpy as np
    sizes = np.asarray(room_sizes, dtype=np.float64)
    rs = np.random.RandomState(_mix(seed, epoch, step, rank) & 0xFFFFFFFF)
    rooms = rs.choice(len(sizes), size=batch_size, p=sizes / sizes.sum())
    counts = np.bincount(rooms, minlength=len(sizes))
    seeds = [_mix(seed, epoch, step, rank, r + 1) for r in range(len(sizes))]
    return counts, seeds


def draw_batch(samplers, batch_size, seed, epoch, step, rank=0):
    """One training batch from the device samplers (one scene.DeviceBlockSampler per room, or one scene.MultiRoomSampler
    over all of them: a single launch per batch) by batch_plan().
    -> (blocks [B,C,N] channel-first view like the loop's points.transpose(2, 1), labels [B,N])"""
    if hasattr(samplers, "table"):                              # scene.MultiRoomSampler
        import numpy as np
        counts, seeds = batch_plan(samplers.sizes, batch_size, seed, epoch, step, rank)
        f, l, info = samplers.sample(np.repeat(np.arange(len(counts)), counts), seeds[0])
        _log_draw(samplers, info)
        return f.permute(0, 2, 1), l
    counts, seeds = batch_plan([sp.P for sp in samplers], batch_size, seed, epoch, step, rank)
    feats, labels, infos = [], [], []
    for sp, n, sd in zip(samplers, counts, seeds):
        if n:
            f, l, info = sp.sample(int(n), seed=sd)
            feats.append(f)
            labels.append(l)
            infos.append(info)
    f = feats[0] if len(feats) == 1 else torch.cat(feats)
    l = labels[0] if len(labels) == 1 else torch.cat(labels)
    _log_draw(samplers[0], infos[0] if len(infos) == 1 else torch.cat(infos))
    return f.permute(0, 2, 1), l


def _log_draw(owner, info):
    """info rows [centre, population, attempts, gave-up] of EVERY draw since the owner's log was last cleared, kept on the
    sampler object (a few hundred bytes per batch, no launch, no host sync): train_epoch reads them once, at the epoch's
    end, where it syncs with the host anyway.  A block whose column never reached the reference's > 1024 points within
    256 attempts comes back as zeros with the flag set -- the reference's loop would spin forever on such a room -- and
    must not be trained on silently, whichever batch of the epoch it was in."""
    log = owner.__dict__.setdefault("draw_log", [])
    log.append(info)
    if len(log) > 65536:                                   # nobody is reading: keep the newest
        del log[:32768]


def gave_up_blocks(samplers, clear=True):
    """Number of blocks the sampler gave up on since the log was last cleared (one host sync)."""
    owner = samplers if hasattr(samplers, "table") else samplers[0]
    log = owner.__dict__.get("draw_log") or []
    n = int(torch.cat([i[:, 3] for i in log]).sum().item()) if log else 0
    if clear:
        del log[:]
    return n


def train_epoch(trainer, samplers, epoch, steps, batch_size, seed=0, learning_rate=1e-3, lr_decay=0.7, step_size=10, rank=None):
    """One epoch of the reference's training loop (modelTraining, localfunctions.py:184-227) with every per-step piece
    on the device: blocks drawn by scene.DeviceBlockSampler (one per room; every block's room drawn in proportion to the
    rooms' point counts, batch_plan()), rotate-z inside the input kernel and accuracy counters on the device (construct
    the trainer with augment=True, metrics=True), the epoch's learning rate and BatchNorm momentum applied graph-safely.
    The next batch is drawn before the current one is stepped (ONE sampler launch per batch whatever the number of rooms:
    scene.MultiRoomSampler), so the trainer's geometry prefetch runs on it meanwhile.  (Drawing two steps ahead on a side
    stream was measured and dropped: work beside the captured step costs its chain more than the 0.1 ms in front of it.)
    rank (default: this process's rank in the trainer's group) enters the sampling seed: replicas train on different
    blocks.  -> {"loss": mean loss, "accuracy": ..., "lr": ..., "bn_momentum": ...} (one host sync at the end)."""
    lr, momentum = epoch_schedule(epoch, learning_rate, lr_decay, step_size)
    trainer.set_lr(lr)
    trainer.set_bn_momentum(momentum)
    if trainer.metrics is not None:
        trainer.metrics.reset()
    if rank is None:
        rank = dist.get_rank(trainer.group) if (dist.is_available() and dist.is_initialized()) else 0

    if not hasattr(samplers, "table") and len(samplers) > 1 and samplers[0].dev.type == "cuda":
        from .scene import MultiRoomSampler
        samplers = MultiRoomSampler(samplers)                  # one launch per batch instead of one per contributing room
    gave_up_blocks(samplers)                               # a fresh log for this epoch
    nxt = draw_batch(samplers, batch_size, seed, epoch, 0, rank)
    loss_sum = None
    for i in range(steps):
        cur, nxt = nxt, draw_batch(samplers, batch_size, seed, epoch, i + 1, rank) if i + 1 < steps else None
        loss = trainer.step(cur[0], cur[1], None if nxt is None else nxt[0])
        loss_sum = loss.clone() if loss_sum is None else loss_sum + loss
    trainer.drop_prefetched()                              # the last batch was announced as its own successor
    out = {"loss": float(loss_sum) / max(steps, 1), "lr": lr, "bn_momentum": momentum}
    gave_up = gave_up_blocks(samplers)
    if gave_up:
        raise RuntimeError("the block sampler gave up on %d block(s) of this epoch: no 1 m column with more than 1024 "
                           "points was found in 256 attempts (a room that sparse makes the reference's sampling loop spin "
                           "forever, sem_seg_training.py:207-216)" % gave_up)
    if trainer.metrics is not None:
        out.update(trainer.metrics.read())
    return out


def label_weights(labels_per_room, num_classes, device=None):
    """calculate_labelweights of the reference's datasets (sem_seg_training.py:264-278): histogram of the labels of all
    rooms, normalised, then (max / w) ** (1/3) -- the class weights the loops pass to nll_loss.  The histogram runs on
    the device the labels live on (torch.bincount); a class that never occurs gets inf, like the reference's division.
    -> float32 tensor [num_classes]"""
    total = None
    for lab in labels_per_room:
        lab = torch.as_tensor(lab)
        if device is not None:
            lab = lab.to(device)
        h = torch.bincount(lab.reshape(-1).to(torch.int64), minlength=num_classes)[:num_classes]
        total = h if total is None else total + h
    w = total.to(torch.float32)
    w = w / w.sum()
    return torch.pow(w.max() / w, 1.0 / 3.0)


class BestModel:
    """The reference's best-model rule (localfunctions.py:310-322): keep the state whenever the evaluation mIoU is at
    least the best so far -- in memory, and as the reference's `best_model.pth` when `path` is given (save_checkpoint)."""

    def __init__(self, path=None):
        self.best_iou = 0.0
        self.epoch = None
        self.state = None
        self.path = path

    def update(self, epoch, miou, model, optimizer=None):
        if miou >= self.best_iou:
            self.best_iou, self.epoch = float(miou), epoch
            self.state = {k: v.detach().clone() for k, v in model.state_dict().items()}
            if self.path is not None:
                save_checkpoint(self.path, epoch, model, optimizer, class_avg_iou=miou)
            return True
        return False


def save_checkpoint(path, epoch, model, optimizer=None, class_avg_iou=None):
    """The reference's checkpoint files (localfunctions.py:229-239 `model.pth` every fifth epoch; :310-322 `best_model.pth`
    with 'class_avg_iou'): {'epoch', ['class_avg_iou',] 'model_state_dict', 'optimizer_state_dict'} through torch.save.
    The model's keys are the reference's (same submodule names), so sem_seg_testing.py:496-497 loads the file as it is;
    optimizer = a SemSegTrainer, a FlatAdam or a torch optimizer."""
    opt = getattr(optimizer, "flat_adam", None) or getattr(optimizer, "optimizer", None) or optimizer
    state = {"epoch": int(epoch), "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}
    if class_avg_iou is not None:
        state["class_avg_iou"] = float(class_avg_iou)
    if opt is not None:
        osd = opt.state_dict()
        state["optimizer_state_dict"] = {"state": {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                                                   for i, st in osd["state"].items()}, "param_groups": osd["param_groups"]}
    torch.save(state, path)
    return state


def load_checkpoint(path, model, optimizer=None):
    """Resume from a checkpoint of save_checkpoint() or of the reference loop (sem_seg_training.py:567-570 reads 'epoch' and
    'model_state_dict'): parameters and buffers are copied IN PLACE (captured graphs stay valid), optimizer moments too.
    -> the checkpoint dict (its 'epoch' is where the reference resumes)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model_state_dict"])
    mlp.invalidate_eval_coefficients()
    opt = getattr(optimizer, "flat_adam", None) or getattr(optimizer, "optimizer", None) or optimizer
    if opt is not None and "optimizer_state_dict" in ck:
        opt.load_state_dict(ck["optimizer_state_dict"])
    return ck


def eval_epoch(model, batches, class_weight=None, batch_size=None, engine=None, metrics=None):
    """The per-epoch evaluation of the reference loop (localfunctions.py:243-308): eval mode, no gradients, over
    `batches` (an iterable of (blocks [b,C,N] channel-first, target [b,N])): mean nll_loss(weight) over the batches,
    accuracy, per-class seen / correct / union counters, mIoU = mean(correct / (union + 1e-6)) (:283), average class
    accuracy (:287-288).  The counters stay on the device (ops.SegMetrics, one kernel per batch) and are read once.
    engine = a scene.BlockInferencer of the model (fixed sub-batch shape): the forward passes are replays of ONE captured
    graph with the next sub-batch's geometry prefetched; its eval coefficients are refreshed from the current weights
    first (mlp.refresh_eval_coefficients).  -> dict like SegMetrics.read() plus "loss"."""
    from .ops import SegMetrics
    from .models.pointnet2_sem_seg import get_loss
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    crit = get_loss()
    batches = list(batches)
    if not batches:
        raise ValueError("eval_epoch: no batches")
    ncls = model.conv2.out_channels
    m = metrics if metrics is not None else SegMetrics(ncls, dev)
    m.reset()
    loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
    with torch.no_grad():
        if engine is not None:
            mlp.refresh_eval_coefficients(model)

            def consume(i, logp):
                tgt = batches[i][1].to(dev)
                nonlocal loss_sum
                loss_sum = loss_sum + crit(logp.reshape(-1, ncls), tgt.reshape(-1), None, class_weight)
                m.add(logp, tgt)
            engine.run([b[0] for b in batches], consume)
        else:
            for x, tgt in batches:
                x, tgt = x.to(dev), tgt.to(dev)
                logp, _ = model(x)
                loss_sum = loss_sum + crit(logp.reshape(-1, ncls), tgt.reshape(-1), None, class_weight)
                m.add(logp, tgt)
    out = m.read()
    out["loss"] = float(loss_sum) / len(batches)            # loss_sum / num_batches (:285)
    if was_training:
        model.train()
    return out
