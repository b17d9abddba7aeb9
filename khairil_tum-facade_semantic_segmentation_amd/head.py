"""Tail of the segmentation head and the loss on the HIP kernels of csrc/pn2_head.hip.

  head_logits(y, conv2)  = F.log_softmax(conv2(y))          reference models/pointnet2_sem_seg.py:37-38
  nll_loss(pred, target, weight)                            reference models/pointnet2_sem_seg.py:44-50

Rows are per-point ([M, K] features -> [M, C] log-probabilities); torch supplies memory and
autograd bookkeeping only."""
import os

import torch

from . import _lib
from .ops import _after_fault_op, _dev, _err_word, _ptr, _stream

IGNORE_INDEX = -100          # F.nll_loss default, which the reference does not override


class _HeadLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w, b, drop_p, seed):
        dev = _dev(y, w, b)
        lib = _lib.load()
        M, K = y.shape
        C = w.shape[0]
        w2 = w.reshape(C, K).contiguous()
        logp = torch.empty((M, C), dtype=torch.float32, device=dev)
        state = _dropout_state(dev) if (seed is None and drop_p > 0.0) else None
        with torch.cuda.device(dev):
            if state is not None:
                # the seed is hashed from a device counter inside the launch and handed to the backward in `seed`
                seed = torch.empty(1, dtype=torch.int64, device=dev)
                rc = lib.pn2_head_logits_dropout_counted(_ptr(y), y.stride(0), _ptr(w2), _ptr(b), _ptr(logp), M, K, C, _ptr(state),
                                                         _ptr(seed), float(drop_p), _stream(dev))
            else:
                rc = lib.pn2_head_logits_dropout(_ptr(y), y.stride(0), _ptr(w2), _ptr(b), _ptr(logp), M, K, C, _ptr(seed),
                                                 float(drop_p), _stream(dev))
        _lib.check(rc, "pn2_head_logits_dropout")
        ctx.save_for_backward(y, w2, logp)
        ctx.drop_p, ctx.seed = float(drop_p), seed
        ctx.wshape = w.shape
        ctx.has_bias = b is not None
        return logp

    @staticmethod
    def backward(ctx, g):
        y, w2, logp = ctx.saved_tensors
        dev = g.device
        lib = _lib.load()
        M, K = y.shape
        C = w2.shape[0]
        g = g.to(torch.float32).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        gy = torch.empty((M, K), **f32) if ctx.needs_input_grad[0] else None
        P = lib.pn2_head_logits_partials(M)
        part = torch.empty((P, C, K + 1), **f32)
        dw = torch.empty((C, K), **f32)
        db = torch.empty(C, **f32) if ctx.has_bias else None
        with torch.cuda.device(dev):
            rc = lib.pn2_head_logits_dropout_backward(_ptr(g), _ptr(logp), _ptr(y), y.stride(0), _ptr(w2), _ptr(gy),
                                                      0 if gy is None else gy.stride(0), _ptr(part), None, None,
                                                      M, K, C, _ptr(ctx.seed), ctx.drop_p, _stream(dev))
        _lib.check(rc, "pn2_head_logits_dropout_backward")
        from . import mlp
        mlp.finish_slabs(part, P, C, K, K, dw, db)       # summed now, or with the stacks' bottom layers at the end of backward
        return gy, dw.view(ctx.wshape), None if db is None else db.view(-1), None, None


def head_logits(y, weight, bias, drop_p=0.0, seed=None):
    """y [M,K] rows, weight [C,K] or [C,K,1] (a 1x1 Conv1d's), bias [C] or None -> log-probs [M,C].
    drop_p > 0: the reference's nn.Dropout(drop_p) in front of conv2 (models/pointnet2_sem_seg.py:36) applied on the fly;
    seed = a one-element int64 tensor on the device (drawn here from torch's generator when omitted), from which
    forward and backward regenerate the keep-mask."""
    dev = _dev(y, weight, bias)
    y = y.to(torch.float32)
    if y.stride(-1) != 1:
        y = y.contiguous()
    if drop_p > 0.0 and seed is None and (not _COUNTED_DROPOUT or _dropout_state(dev) is None):
        seed = torch.randint(-2 ** 62, 2 ** 62, (1,), dtype=torch.int64, device=dev)
    if drop_p <= 0.0:
        seed = None
    return _HeadLogits.apply(y, weight, bias, float(drop_p), seed)


def dropout_mask(seed, drop_p, M, K):
    """The keep-mask head_logits(..., drop_p, seed) applies to y [M,K] (bool tensor); for tests."""
    dev = seed.device
    lib = _lib.load()
    mask = torch.empty((M, K), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_dropout_mask(_ptr(seed), float(drop_p), M, K, _ptr(mask), _stream(dev))
    _lib.check(rc, "pn2_dropout_mask")
    return mask.bool()


# The head's dropout seed from a device counter (pn2_head_logits_dropout_counted): no random-number launch in front of the head,
# and a captured training step draws nothing from torch's generator -- replaying a graph that does costs two fill launches for
# the generator state on top (14 us per step together).  The base seed comes from torch's generator once per device, so
# torch.manual_seed() still decides the masks.  PN2_COUNTED_DROPOUT=0: a torch.randint per call.
_COUNTED_DROPOUT = os.environ.get("PN2_COUNTED_DROPOUT", "1") == "1"
_dropout_states = {}


def _dropout_state(dev):
    """[base seed, calls so far, ticket, unused] on the device; None while a stream capture is running and the state does not
    exist yet (ensure_ticket_words() before the capture creates it)."""
    if not _COUNTED_DROPOUT:
        return None
    key = (dev.type, dev.index)
    st = _dropout_states.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        st = torch.zeros(4, dtype=torch.int64, device=dev)
        st[0:1] = torch.randint(-2 ** 62, 2 ** 62, (1,), dtype=torch.int64, device=dev)
        _dropout_states[key] = st
    return st


_tickets = {}
# PN2_NLL_TICKET=0: the two-launch form of the loss (partials, then a finalize launch) for A/B runs
_NLL_TICKET = os.environ.get("PN2_NLL_TICKET", "1") == "1"


def _ticket_word(dev):
    """The zeroed device word pn2_nll_loss_ticketed counts finished workgroups in (the kernel returns it to zero).
    Launches that share a word must be ordered: one word per (device, stream) for eager launches, one per device for
    launches under stream capture (a graph orders its own nodes; two graphs replaying loss kernels of one device at
    the same time would need ensure_ticket_words() and their own words -- not a case this package creates)."""
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev.type, dev.index, "capture" if capturing else torch.cuda.current_stream(dev).cuda_stream)
    word = _tickets.get(key)
    if word is None:
        if capturing:
            # a word created now would live in the graph's pool and be zero-filled by every replay (one more launch):
            # still correct; ensure_ticket_words() before the capture avoids it
            return torch.zeros(1, dtype=torch.int32, device=dev)
        word = _tickets[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return word


def ensure_ticket_words(dev):
    """Create the device's ticket word for captured launches (call before capturing a graph that holds a loss)."""
    key = (dev.type, dev.index, "capture")
    if key not in _tickets and not torch.cuda.is_current_stream_capturing():
        _tickets[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    _dropout_state(dev)


class _NLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target, weight):
        dev = _dev(logp, target, weight)
        lib = _lib.load()
        M, C = logp.shape
        P = lib.pn2_nll_loss_partials(M)
        part = torch.empty((P, 2), dtype=torch.float64, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)           # loss, sum of weights
        with torch.cuda.device(dev):
            if not _NLL_TICKET:
                rc = lib.pn2_nll_loss(_ptr(logp), _ptr(target), _ptr(weight), M, C, IGNORE_INDEX, _ptr(part), _ptr(out),
                                      out.data_ptr() + 4, _ptr(_err_word(dev)), _stream(dev))
                _lib.check(rc, "pn2_nll_loss")
        with torch.cuda.device(dev):
            rc = 0 if not _NLL_TICKET else lib.pn2_nll_loss_ticketed(_ptr(logp), _ptr(target), _ptr(weight), M, C, IGNORE_INDEX, _ptr(part), _ptr(out),
                                           out.data_ptr() + 4, _ptr(_err_word(dev)), _ptr(_ticket_word(dev)), _stream(dev))
        _lib.check(rc, "pn2_nll_loss_ticketed")
        _after_fault_op(dev, "nll_loss")
        ctx.save_for_backward(target, weight, out)
        ctx.shape = (M, C)
        return out[0]

    @staticmethod
    def backward(ctx, gl):
        target, weight, out = ctx.saved_tensors
        dev = gl.device
        lib = _lib.load()
        M, C = ctx.shape
        gl = gl.to(torch.float32).contiguous()
        glogp = torch.empty((M, C), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.pn2_nll_loss_backward(_ptr(gl), _ptr(target), _ptr(weight), out.data_ptr() + 4, M, C, IGNORE_INDEX,
                                           _ptr(glogp), _stream(dev))
        _lib.check(rc, "pn2_nll_loss_backward")
        return glogp, None, None


def nll_loss(pred, target, weight=None):
    """pred [M,C] log-probabilities, target [M] int64, weight [C] or None -> scalar mean loss."""
    _dev(pred, target, weight)
    if pred.dim() != 2 or target.dim() != 1 or target.shape[0] != pred.shape[0]:
        raise ValueError("nll_loss expects pred [M,C] and target [M], got %s and %s" % (tuple(pred.shape), tuple(target.shape)))
    pred = pred.to(torch.float32).contiguous()
    target = target.to(torch.int64).contiguous()
    if weight is not None:
        if weight.numel() != pred.shape[1]:
            raise ValueError("weight has %d entries for %d classes" % (weight.numel(), pred.shape[1]))
        weight = weight.detach().to(torch.float32).contiguous()
    return _NLL.apply(pred, target, weight)
