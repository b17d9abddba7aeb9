"""Per-module hipGraph replay for the drop-in surface (models/pointnet2_utils.py).

A caller who keeps the reference's models/pointnet2_sem_seg.py:22-40 and the loop of localfunctions.py:203-218 calls eight
modules per step, eagerly: about 250 kernel launches whose ENQUEUE (ctypes, torch.empty, autograd bookkeeping: ~20 us each)
takes longer than the kernels run.  Nothing in the caller can be changed, so the modules themselves turn repetition into
replay: after WARMUP eager calls with one signature (shapes, strides, mode, which tensors want gradients, where the
parameters live) a module captures its forward -- and its backward, torch.cuda.make_graphed_callables' pattern: static
inputs, the autograd graph of the captured forward differentiated under a second capture into static gradients -- and every
later call with that signature is one (forward) + one (backward) graph launch behind a torch.autograd.Function.

What make_graphed_callables leaves to the caller is handled here, because this caller does not know it is being graphed:

* Outputs are static buffers, overwritten by the next replay.  An instance is only replayed while nobody can still observe
  (or needs) what its last replay produced: the storage use-count of every static output is back to ours alone, and the
  autograd graph of its last call has released its saved variables (a weak reference to a marker tensor saved for the
  backward dies exactly then -- after backward() without retain_graph, or when the graph is dropped).  Otherwise another
  instance of the same signature runs (captured on demand, at most MAX_INSTANCES), else the call is eager.  The reference
  loop keeps `trans_feat` (the last set-abstraction output) bound across one iteration: that module alternates between
  two instances and everything downstream of it follows through its keys, with no copy anywhere.
* Chaining without copies: a static output handed to the next module arrives with the same storage address every time.  A
  module whose input IS another instance's static buffer captures that address directly (the signature carries it) instead
  of copying into a buffer of its own; inputs from anywhere else are copied in.  The inputs of a call are saved for its
  backward, so the producer cannot be replayed over them before the consumer's backward has read them.
* Parameter gradients leave the backward graph as static buffers, which autograd's AccumulateGrad adopts as `.grad` when
  it is None.  If a parameter's `.grad` still aliases the buffer when the instance is about to be replayed again
  (gradient accumulation, zero_grad(set_to_none=False)), it is detached into a copy first -- before the FORWARD replay
  already: inside an instance's pool a gradient buffer may occupy a block the forward uses for an intermediate.
* FPS start indices are an INPUT of the captured forward (drawn -- or taken from fps_starts() -- outside the graph with
  the reference's own torch.randint call, pointnet2_utils.py:75), BatchNorm momentum is the device word of mlp.py, eval
  coefficients are refreshed in place before an eval replay: a replay computes what the eager call would.
* Every instance owns its memory pool: instances are captured in call order (forward and backward of one module
  together) but replayed forward-chain-then-backward-chain, so a pool shared across modules would let one module's
  forward reuse blocks another's pending backward still reads.

PN2_MODULE_GRAPHS=0 switches the whole mechanism off (eager modules, the round-3 behaviour)."""
import os
import weakref

import torch

from . import mlp, ops

ENABLED = os.environ.get("PN2_MODULE_GRAPHS", "1") == "1"
WARMUP = int(os.environ.get("PN2_MODULE_GRAPHS_WARMUP", "2"))    # eager calls of a signature before its capture
_DEBUG = os.environ.get("PN2_MODULE_GRAPHS_DEBUG", "0") == "1"
MAX_INSTANCES = 3                                                 # per full signature
MAX_SIGNATURES = 12                                               # per module (shapes x modes x input addresses)

stats = {"captures": 0, "replays": 0, "eager": 0, "busy": 0, "backward_replays": 0, "grad_detached": 0}
_STATIC = {}            # storage address -> True for every live static output buffer (what "arrives at the same address")


def _use_count(t):
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


_BASE_USE = None        # use-count of a storage held by exactly one tensor, as _use_count() sees it


def _base_use():
    global _BASE_USE
    if _BASE_USE is None:
        _BASE_USE = _use_count(torch.empty(1))
    return _BASE_USE


def usable(*tensors):
    """Can this call go through a graph at all?"""
    if not ENABLED or mlp._GATE_TAPS is not None or ops._ERROR_MODE != "lazy":
        return False
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or (dev is not None and t.device != dev):
            return False
        dev = t.device
    if dev is None or torch.cuda.is_current_stream_capturing() or torch.is_autocast_enabled():
        return False
    return True


def _slots(module):
    """([(owner module, dict name, key)] of every parameter and buffer below `module`, [BatchNorm modules]), resolved once;
    the tensors are looked up through the owners on every call, so a replaced Parameter is seen."""
    hit = module.__dict__.get("_pn2_graph_slots")
    if hit is None:
        slots, bns = [], []
        for m in module.modules():
            for k in m._parameters:
                slots.append((m, "_parameters", k))
            for k in m._buffers:
                slots.append((m, "_buffers", k))
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                bns.append(m)
        hit = module.__dict__["_pn2_graph_slots"] = (slots, bns)
    return hit


def _state(module):
    """(parameters, key): the key holds what a captured graph has baked in -- where every parameter and buffer lives, which
    parameters want a gradient, and per BatchNorm the mode, eps and whether momentum is a number (its VALUE is a device
    word, mlp.momentum_word)."""
    slots, bns = _slots(module)
    params, key = [], []
    for m, d, k in slots:
        t = getattr(m, d)[k]
        if t is None:
            key.append(None)
            continue
        key.append((t.data_ptr(), t.requires_grad))
        if d == "_parameters":
            params.append(t)
    for bn in bns:
        key.append((bn.training, bn.eps, bn.momentum is None, bn.track_running_stats))
    return params, tuple(key)


def _tensor_sig(t):
    if t is None:
        return None, None
    shape = (tuple(t.shape), tuple(t.stride()), t.dtype, t.requires_grad)
    base = t.untyped_storage().data_ptr()
    return shape, ((base, t.storage_offset()) if base in _STATIC else None)


class _Instance:
    def __init__(self):
        self.fwd = self.bwd = None
        self.own_inputs = []          # per input: our static buffer (copied into before a replay) or None (captured address)
        self.static_outputs = []
        self.out_requires_grad = []
        self.static_grad_outputs = [] # per output (None where no gradient flows)
        self.static_grad_inputs = []  # per (input..., parameter...) position
        self.param_grad_ptrs = []
        self.params = []
        self.pending = None           # weakref to the marker saved for the backward of the last call
        self.n_in = 0

    def free(self):
        if self.pending is not None and self.pending() is not None:
            return False
        base = _base_use()
        for o in self.static_outputs:
            if _use_count(o) > base:
                return False
        return True

    def copy_in(self, inputs):
        for own, t in zip(self.own_inputs, inputs):
            if own is not None:
                own.copy_(t)

    def detach_adopted_gradients(self):
        """A parameter whose .grad still IS one of our gradient buffers (adopted by AccumulateGrad last time and not reset
        to None since) gets a copy of its own.  Called before EVERY replay, the forward's too: the buffers live in this
        instance's pool, where a block the backward capture took for a gradient may be one the forward capture had used
        for an intermediate and freed -- a forward replay then scribbles over the adopted gradient."""
        for p, ptr in zip(self.params, self.param_grad_ptrs):
            if ptr is not None and p.grad is not None and p.grad.untyped_storage().data_ptr() == ptr:
                p.grad = p.grad.clone()
                stats["grad_detached"] += 1

    def release(self):
        for o in self.static_outputs:
            _STATIC.pop(o.untyped_storage().data_ptr(), None)


class _Replay(torch.autograd.Function):
    """One forward replay now, one backward replay when (if) the gradient arrives."""

    @staticmethod
    def forward(ctx, inst, *args):
        inputs = args[:inst.n_in]
        inst.detach_adopted_gradients()
        inst.copy_in(inputs)
        inst.fwd.replay()
        stats["replays"] += 1
        marker = torch.empty(0)
        inst.pending = weakref.ref(marker)
        ctx.inst = inst
        # the marker lives exactly as long as this call's backward may still run; the inputs are held for as long, so the
        # instance that produced them (their static buffers are read by OUR backward graph) is not replayed before that
        ctx.save_for_backward(marker, *[t for t in inputs if t is not None])
        outs = tuple(o.detach() for o in inst.static_outputs)
        nondiff = [o for o, r in zip(outs, inst.out_requires_grad) if not r]
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        inst = ctx.inst
        ctx.saved_tensors                       # raises the usual error on a second backward through a freed graph
        for sg, g in zip(inst.static_grad_outputs, gouts):
            if sg is None:
                continue
            if g is None:
                sg.zero_()
            elif g.data_ptr() != sg.data_ptr() or g.stride() != sg.stride():
                sg.copy_(g)
        inst.detach_adopted_gradients()
        inst.bwd.replay()
        stats["backward_replays"] += 1
        return (None,) + tuple(None if s is None else s.detach() for s in inst.static_grad_inputs)


class _leaf_parameters:
    """While the block is open every parameter below `module` is a fresh leaf ALIAS (same memory, no autograd history).
    The capture differentiates with respect to these: the real parameters' AccumulateGrad nodes were created by the
    caller's earlier eager passes on the caller's stream and stay alive as long as the caller keeps any loss around (the
    reference loop sums them: localfunctions.py:221), and routing a gradient to such a node makes autograd synchronise
    ITS stream with the capturing one -- which pulls the default stream into the capture and ends in a crash at
    hipStreamEndCapture (seen on MI355X).  An alias gets its accumulator on the capture stream."""

    def __init__(self, module):
        self.module = module
        self.saved = []
        self.aliases = []

    def __enter__(self):
        for m, d, k in _slots(self.module)[0]:
            if d != "_parameters" or m._parameters[k] is None:
                continue
            p = m._parameters[k]
            a = p.detach().requires_grad_(p.requires_grad)
            self.saved.append((m, k, p))
            self.aliases.append(a)
            m._parameters[k] = a
        return self.aliases

    def __exit__(self, *exc):
        for m, k, p in self.saved:
            m._parameters[k] = p
        return False


def _capture(module, fn, inputs, stable, params, want_backward):
    """Capture fn(*inputs) (and its backward) into a new instance.  inputs whose `stable` entry is set are other instances'
    static buffers: their addresses go into the graph as they are."""
    inst = _Instance()
    inst.n_in = len(inputs)
    inst.params = list(params)
    static_in = []
    for t, st in zip(inputs, stable):
        if t is None:
            inst.own_inputs.append(None)
            static_in.append(None)
        elif st is not None:
            inst.own_inputs.append(None)
            static_in.append(t.detach().requires_grad_(t.requires_grad))     # an alias for the duration of the capture only
        else:
            own = torch.empty_like(t).copy_(t)
            inst.own_inputs.append(own)
            static_in.append(own.detach().requires_grad_(t.requires_grad))
    dev = next(t.device for t in inputs if t is not None)
    pool = torch.cuda.graph_pool_handle()
    if _DEBUG:
        print("[graphed] capture forward of", getattr(fn, "__self__", fn).__class__.__name__, [None if t is None else tuple(t.shape) for t in inputs],
              "backward" if want_backward else "", flush=True)
    inst.fwd = torch.cuda.CUDAGraph()
    with _leaf_parameters(module) as leaves:
        with torch.cuda.graph(inst.fwd, pool=pool):
            with torch.enable_grad() if want_backward else torch.no_grad():
                outs = fn(*static_in)
    outs = tuple(outs)
    assert len(leaves) == len(params) and all(a.data_ptr() == p.data_ptr() for a, p in zip(leaves, params))
    in_bases = {t.untyped_storage().data_ptr() for t in static_in if t is not None}
    if any(o.untyped_storage().data_ptr() in in_bases for o in outs):
        raise RuntimeError("graphed module: an output aliases an input")
    inst.out_requires_grad = [bool(o.requires_grad) for o in outs]
    if want_backward and any(inst.out_requires_grad):
        diff_out = [o for o in outs if o.requires_grad]
        inst.static_grad_outputs = [torch.empty_like(o) if o.requires_grad else None for o in outs]
        wrt = [t for t in static_in if t is not None and t.requires_grad] + [a for a in leaves if a.requires_grad]
        inst.bwd = torch.cuda.CUDAGraph()
        if _DEBUG:
            print("[graphed] capture backward", len(diff_out), "outputs,", len(wrt), "gradients", flush=True)
        with torch.cuda.graph(inst.bwd, pool=pool):
            grads = torch.autograd.grad(diff_out, wrt, grad_outputs=[g for g in inst.static_grad_outputs if g is not None],
                                        allow_unused=True)
        if _DEBUG:
            print("[graphed] backward captured", flush=True)
        grads = list(grads)
        for t in static_in:
            inst.static_grad_inputs.append(grads.pop(0) if (t is not None and t.requires_grad) else None)
        for p in params:
            g = grads.pop(0) if p.requires_grad else None
            inst.static_grad_inputs.append(g)
            inst.param_grad_ptrs.append(None if g is None else g.untyped_storage().data_ptr())
    else:
        inst.static_grad_outputs = [None] * len(outs)
        inst.static_grad_inputs = [None] * (len(static_in) + len(params))
        inst.param_grad_ptrs = [None] * len(params)
    inst.static_outputs = [o.detach() for o in outs]
    del outs, static_in, leaves
    for o in inst.static_outputs:
        ptr = o.untyped_storage().data_ptr()
        _STATIC[ptr] = True
        weakref.finalize(inst, _STATIC.pop, ptr, None)        # an address of a dead instance means nothing
    stats["captures"] += 1
    torch.cuda.current_stream(dev).synchronize()
    return inst


class _ModuleGraphs:
    def __init__(self):
        self.seen = {}            # signature without addresses -> eager calls so far
        self.instances = {}       # full signature -> [instances]
        self.refused = set()

    def drop(self):
        for lst in self.instances.values():
            for inst in lst:
                inst.release()
        self.instances.clear()


def _book(module):
    book = module.__dict__.get("_pn2_graphs")
    if book is None:
        book = module.__dict__["_pn2_graphs"] = _ModuleGraphs()
    return book


def reset(module=None):
    """Forget every captured graph (of `module` and its children, or -- None -- nothing global is kept)."""
    if module is not None:
        for m in module.modules():
            book = m.__dict__.pop("_pn2_graphs", None)
            if book is not None:
                book.drop()
            m.__dict__.pop("_pn2_graph_slots", None)


def call(module, fn, inputs, before_replay=None):
    """fn(*inputs) through a captured graph when there is one, eagerly otherwise.  inputs: tensors or None; fn returns a
    tuple of tensors.  before_replay(): host-side refreshes a replay needs (momentum words, eval coefficients)."""
    book = _book(module)
    params, pkey = _state(module)
    grad_mode = torch.is_grad_enabled()
    shapes, stable = zip(*[_tensor_sig(t) for t in inputs])
    short = (module.training, grad_mode, shapes, pkey)
    n = book.seen.get(short, 0)
    if n < WARMUP or short in book.refused:
        book.seen[short] = n + 1
        stats["eager"] += 1
        return fn(*inputs)
    full = (short, stable)
    lst = book.instances.get(full)
    if lst is None:
        if len(book.instances) >= MAX_SIGNATURES:
            book.drop()                                    # shapes keep changing: start over rather than grow
        lst = book.instances[full] = []
    inst = None
    for cand in lst:
        if cand.free():
            inst = cand
            break
    want_backward = grad_mode and (any(t is not None and t.requires_grad for t in inputs) or any(p.requires_grad for p in params))
    if before_replay is not None:
        before_replay()
    if inst is None:
        if len(lst) >= MAX_INSTANCES:
            stats["busy"] += 1
            stats["eager"] += 1
            return fn(*inputs)
        try:
            from . import ops
            with ops.capture_region():                  # no garbage collection inside a capture (ops.capture_region)
                inst = _capture(module, fn, inputs, stable, params, want_backward)
        except RuntimeError as e:
            # a capture launches nothing and has no side effect on the model (parameters are swapped back, counters are
            # bumped by captured kernels only): whatever made it fail, the eager call is still right.  The signature is
            # not tried again; anything but the expected refusal is reported once.
            if "aliases an input" not in str(e):
                import warnings
                warnings.warn("graphed module: capture of %s failed (%s); this signature stays eager"
                              % (type(module).__name__, str(e).splitlines()[0][:200]))
            book.refused.add(short)
            stats["eager"] += 1
            return fn(*inputs)
        lst.append(inst)
    if inst.bwd is not None:
        return _Replay.apply(inst, *inputs, *params)
    inst.copy_in(inputs)
    inst.fwd.replay()
    stats["replays"] += 1
    inst.pending = None
    return tuple(o.detach() for o in inst.static_outputs)
