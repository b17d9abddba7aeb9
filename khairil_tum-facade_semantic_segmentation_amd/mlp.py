"""Fused [Conv 1x1 -> BatchNorm -> ReLU] x n (+ max over nsample) stacks on the MFMA kernels of
csrc/pn2_mlp.hip (C ABI: pn2_mlp_gemm / pn2_bn_* / pn2_mlp_dw in include/pn2_hip.h).

Reference semantics: models/pointnet2_utils.py:196-200 (set abstraction) and :312-314 (feature
propagation) with nn.BatchNorm2d/1d in train or eval mode.  torch supplies memory and autograd
bookkeeping only; every FLOP of the stack runs in the HIP library."""
import os

import torch

from . import _lib
from .ops import _dev, _ptr, _stream

PRO_NONE, PRO_BN_RELU, PRO_BN_BWD = 0, 1, 2

# The weight-gradient kernels of a stack's backward depend only on tensors the dX chain has
# already produced, so they run on a side stream (a parallel branch under hipGraph capture) and
# fill idle CUs.  Measured on MI355X: 5.06 ms/step with it against 4.81 without (the two kernel
# families thrash each other), so it is OFF by default; PN2_DW_SIDE_STREAM=1 enables it.
_DW_SIDE = os.environ.get("PN2_DW_SIDE_STREAM", "0") == "1"
# ... but for stacks with few rows (deep levels: neither the dX GEMM nor the dW kernel fills the 256 CUs) the
# parallel branch does pay; PN2_DW_SIDE_MAX_ROWS is the largest M that uses it (0 = never).
_DW_SIDE_MAX_ROWS = int(os.environ.get("PN2_DW_SIDE_MAX_ROWS", "0"))
# Layers with at most 128 input and output channels run their whole backward (dX, dW, db, statistics for
# the layer below) in one pass over the activations (pn2_mlp_bwd_layer); PN2_FUSED_BWD=0 keeps the
# two-kernel path (pn2_mlp_gemm prologue 2 + pn2_mlp_dw) for A/B runs.
_FUSED_BWD = os.environ.get("PN2_FUSED_BWD", "1") == "1"
# The last layer's BatchNorm finalize and the stack's output (BatchNorm + ReLU, max over nsample = 32 from the extrema
# the GEMM epilogue recorded) run in one launch (pn2_bn_finalize_out); PN2_FUSED_OUT=0 keeps pn2_bn_finalize +
# pn2_bn_relu_out (which re-reads z) for A/B runs.
_FUSED_OUT = os.environ.get("PN2_FUSED_OUT", "1") == "1"
_side_streams = {}
# Test hook (tests/test_hip_gates.py): while a list, every stack forward appends the tensors that define its ReLU /
# max-pool decisions -- references to what the backward keeps anyway, nothing is copied or launched.
_GATE_TAPS = None


class record_gates:
    """with mlp.record_gates() as taps: ... -- taps[i] = {"zs": raw conv outputs per layer, "coefs": (scale, shift, mean,
    invstd) per layer, "argk": winning row of each pooled group (or None), "pool_k": rows per group, "y": the stack's
    output} of the i-th stack that ran, in call order."""

    def __enter__(self):
        global _GATE_TAPS
        self._prev = _GATE_TAPS
        _GATE_TAPS = []
        return _GATE_TAPS

    def __exit__(self, *exc):
        global _GATE_TAPS
        _GATE_TAPS = self._prev
        return False


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _gemm(lib, dev, x1, K1, x2, K2, pro, consts, argk, pool_k, w, ldw, w_is_kn, bias, out, M, N, stat=None,
          mask=None, out2=None, nsplit=0):
    """Thin positional wrapper of pn2_mlp_gemm.  consts = (scale, shift, mean, invstd, c1, c2) or None;
    mask = (mask_z, mscale, mshift, mmean, minvstd) or None."""
    c = consts or (None,) * 6
    m = mask or (None,) * 5
    rc = lib.pn2_mlp_gemm(_ptr(x1), x1.stride(0), K1, _ptr(x2), 0 if x2 is None else x2.stride(0), K2, pro,
                          _ptr(c[0]), _ptr(c[1]), _ptr(c[2]), _ptr(c[3]), _ptr(c[4]), _ptr(c[5]),
                          _ptr(argk), pool_k, _ptr(w), ldw, w_is_kn, _ptr(bias), _ptr(out), out.stride(0),
                          _ptr(out2), 0 if out2 is None else out2.stride(0), nsplit, M, N, _ptr(stat), _ptr(m[0]), 0 if m[0] is None else m[0].stride(0), _ptr(m[1]), _ptr(m[2]),
                          _ptr(m[3]), _ptr(m[4]), _stream(dev))
    _lib.check(rc, "pn2_mlp_gemm")


def momentum_word(bn, dev):
    """(host value, device word) of a BatchNorm module's momentum.  The kernels read the DEVICE word when they run,
    so a step replayed from a hipGraph follows later changes of `bn.momentum` (the reference loop resets it every
    epoch, localfunctions.py:191-195) -- provided the word is refreshed outside the graph: set_bn_momentum() does
    that, and so does every eager forward.  momentum=None (cumulative moving average) has no word: (-1, None)."""
    if bn.momentum is None:
        return -1.0, None
    m = float(bn.momentum)
    word = bn.__dict__.get("_pn2_momentum")
    if word is None or word[0].device != dev:
        if torch.cuda.is_current_stream_capturing():
            # a word created now would live in the graph's pool and be re-filled by every replay: the launch gets
            # the host value instead (frozen).  ensure_momentum_words() before the capture avoids this.
            return m, None
        word = [torch.full((1,), m, dtype=torch.float32, device=dev), m]
        bn.__dict__["_pn2_momentum"] = word
    elif word[1] != m and not torch.cuda.is_current_stream_capturing():
        word[0].fill_(m)
        word[1] = m
    return m, word[0]


def ensure_momentum_words(module):
    """Create the device words of every BatchNorm below `module` (call before capturing a graph)."""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) and m.weight is not None and m.weight.is_cuda:
            momentum_word(m, m.weight.device)


def set_bn_momentum(module, momentum):
    """The reference's per-epoch `m.momentum = ...` for every BatchNorm below `module` (localfunctions.py:187-195),
    graph-safe: the device words the captured bn_finalize launches read are refreshed too."""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.momentum = momentum
            word = m.__dict__.get("_pn2_momentum")
            if word is not None and momentum is not None:
                word[0].fill_(float(momentum))
                word[1] = float(momentum)


# The tensors an eval-mode BatchNorm's coefficients derive from are updated through RAW POINTERS (pn2_bn_finalize* write
# running_mean / running_var, pn2_adam_step* write gamma / beta through FlatAdam.flat, hipGraph replays do both): none of
# that bumps a version counter.  Every such writer bumps this generation instead, and a cached pair is valid for one
# generation only.
_generation = [0]


def invalidate_eval_coefficients():
    """Parameters or BatchNorm buffers were (or may have been) written behind torch's back: cached eval coefficients are
    stale.  Called by every training-mode stack forward, by FlatAdam and by SemSegTrainer.step."""
    _generation[0] += 1


def _eval_coefficients(lib, dev, bn, gamma, beta):
    """scale / shift of an eval-mode BatchNorm.  Computed once per generation (see above) and kept on the module: an
    inference pass over the network launches none of the 22 pn2_bn_eval_coeff kernels again.  A stale pair is recomputed
    INTO THE SAME TENSORS, so a captured graph that reads them (scene.BlockInferencer) sees the refreshed values after
    refresh_eval_coefficients()."""
    key = tuple((t.data_ptr(), t._version) for t in (gamma, beta, bn.running_mean, bn.running_var)) + \
        (float(bn.eps), str(dev), _generation[0])
    hit = bn.__dict__.get("_pn2_eval_coeff")
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    Co = gamma.shape[0]
    capturing = torch.cuda.is_current_stream_capturing()
    if hit is not None and not capturing and hit[1].shape[0] == Co and hit[1].device == dev:
        scale, shift = hit[1], hit[2]
    else:
        scale = torch.empty(Co, dtype=torch.float32, device=dev)
        shift = torch.empty(Co, dtype=torch.float32, device=dev)
    rc = lib.pn2_bn_eval_coeff(Co, _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var),
                               float(bn.eps), _ptr(scale), _ptr(shift), _stream(dev))
    _lib.check(rc, "pn2_bn_eval_coeff")
    if not capturing:                                      # tensors of a graph's pool must not outlive it on a module
        bn.__dict__["_pn2_eval_coeff"] = (key, scale, shift)
    return scale, shift


def refresh_eval_coefficients(module):
    """Recompute (in place) the cached eval coefficients of every BatchNorm below `module` from its current weights and
    running statistics: what a replayed inference graph needs after the model was trained further."""
    lib = _lib.load()
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) and "_pn2_eval_coeff" in m.__dict__ and m.weight is not None:
            dev = m.weight.device
            with torch.cuda.device(dev):
                _eval_coefficients(lib, dev, m, m.weight, m.bias)


_DEFERRED = None            # list of pending slab sums while a deferred_weight_sums() block is open


def reduce_slabs(jobs, dev):
    """jobs: (partial [P,N,K+1], P, N, K, Kstore, dw [N,Kstore], db [N] or None) -> one pn2_mlp_dw_reduce_many launch per 16."""
    import ctypes
    if not jobs:
        return
    n = len(jobs)
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    with torch.cuda.device(dev):
        rc = _lib.load().pn2_mlp_dw_reduce_many(
            n, vp(*[j[0].data_ptr() for j in jobs]), ci(*[j[1] for j in jobs]), ci(*[j[2] for j in jobs]),
            ci(*[j[3] for j in jobs]), ci(*[j[4] for j in jobs]), vp(*[j[5].data_ptr() for j in jobs]),
            vp(*[None if j[6] is None else j[6].data_ptr() for j in jobs]), _stream(dev))
    _lib.check(rc, "pn2_mlp_dw_reduce_many")


class deferred_weight_sums:
    """Inside the block, the weight gradients of the BOTTOM layer of every stack (and of the head's last conv) stay as
    per-workgroup slabs; leaving the block sums all of them in one launch.  Nothing in a backward pass reads those
    gradients -- only the optimizer does -- so a training step wraps `loss.backward()` in this (train.SemSegTrainer): nine
    launches of the step become one.  The gradient TENSORS are handed to autograd at once and filled at the end of the
    block: only valid while every `.grad` is None on entry (autograd then keeps the tensor it is given; an accumulating
    `grad += g` would read it too early), which the block checks for the parameters it is given."""

    def __init__(self, params=None):
        self.params = params

    def __enter__(self):
        global _DEFERRED
        self.outer = _DEFERRED is not None
        if not self.outer:
            if self.params is not None and any(p.grad is not None for p in self.params):
                self.outer = True                        # somebody accumulates: stay with immediate sums
            else:
                _DEFERRED = []
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        if self.outer:
            return False
        jobs, _DEFERRED = _DEFERRED, None
        if exc[0] is None and jobs:
            reduce_slabs(jobs, jobs[0][0].device)
            if self.params is not None:
                # autograd keeps a gradient it is handed only while nobody else holds the same tensor object (backward
                # returns VIEWS of dw / db for that reason); a copy made before the sums would hold unfinished values
                held = {p.grad.data_ptr() for p in self.params if p.grad is not None}
                for j in jobs:
                    if j[5].data_ptr() not in held or (j[6] is not None and j[6].data_ptr() not in held):
                        raise RuntimeError("deferred_weight_sums: autograd copied a weight gradient before it was summed "
                                           "(set PN2_DEFER_DW=0)")
        return False


def finish_slabs(partial, P, N, K, Kstore, dw, db):
    """Sum one layer's slabs now, or at the end of the open deferred_weight_sums() block."""
    job = (partial, P, N, K, Kstore, dw, db)
    if _DEFERRED is not None:
        _DEFERRED.append(job)
    else:
        reduce_slabs([job], partial.device)


class _MLPStack(torch.autograd.Function):
    """y = stack(x1 | x2).  args: bns (list of nn.BatchNorm modules, for running stats / mode),
    pool_k (0 = no pooling), x1 [M,K1], x2 [M,K2] or None, then per layer conv_w, conv_b, bn_w, bn_b.  The first
    conv's weight may have fewer columns than the rows (zero pad columns behind the real ones, 16-byte aligned rows): it is
    padded here, and its gradient comes back in its own shape."""

    @staticmethod
    def forward(ctx, bns, pool_k, x1, x2, *params):
        dev = _dev(x1, x2)
        lib = _lib.load()
        L = len(bns)
        M, K1 = x1.shape
        K2 = 0 if x2 is None else x2.shape[1]
        training = bns[0].training
        # nobody will back-propagate through this call (torch.no_grad() inference): what only the backward reads is not
        # written
        inference = not training and not any(ctx.needs_input_grad)
        f32 = dict(dtype=torch.float32, device=dev)
        zs, coefs = [], []
        y = argk = wpad = None
        with torch.cuda.device(dev):
            for l in range(L):
                w, b, gamma, beta = params[4 * l:4 * l + 4]
                Co = w.shape[0]
                w2 = w.reshape(Co, -1)
                if l == 0 and w2.shape[1] < K1 + K2:
                    w2 = wpad = _pad_cols(w2, K1 + K2)
                z = None if (inference and l == L - 1 and _FUSED_OUT and pool_k == 32 and Co % 4 == 0 and M % 32 == 0) \
                    else torch.empty((M, Co), **f32)
                P = lib.pn2_mlp_gemm_max_partials(M)
                stat = torch.empty((P, 2, Co), **f32) if training else None
                last = l == L - 1
                fuse_out = last and _FUSED_OUT and pool_k in (0, 32) and Co % 4 == 0
                pooled = None
                if l == 0:
                    src = (x1, K1, x2, K2, PRO_NONE, None, None)
                else:
                    zp = zs[-1]
                    src = (zp, zp.shape[1], None, 0, PRO_BN_RELU, coefs[-1][0], coefs[-1][1])
                if fuse_out and pool_k == 32 and M % 32 == 0:
                    # the max over nsample falls out of the epilogue: per (group, channel) the extrema of z
                    pooled = (torch.empty((M // 32, Co), **f32), torch.empty((M // 32, Co), **f32),
                              torch.empty((M // 32, Co), dtype=torch.uint8, device=dev),
                              torch.empty((M // 32, Co), dtype=torch.uint8, device=dev))
                    a1, ak1, a2, ak2, pro, psc, psh = src
                    rc = lib.pn2_mlp_gemm_pool32(_ptr(a1), a1.stride(0), ak1, _ptr(a2), 0 if a2 is None else a2.stride(0), ak2,
                                                 pro, _ptr(psc), _ptr(psh), _ptr(w2), w2.stride(0), _ptr(b), _ptr(z),
                                                 Co if z is None else z.stride(0), M, Co, _ptr(stat), _ptr(pooled[0]),
                                                 _ptr(pooled[1]), _ptr(pooled[2]), _ptr(pooled[3]), _stream(dev))
                    if rc == _lib.ERR_UNSUPPORTED:      # operands the pipelined kernels do not take: nothing was launched
                        pooled = None
                        if z is None:
                            z = torch.empty((M, Co), **f32)
                    else:
                        _lib.check(rc, "pn2_mlp_gemm_pool32")
                if pooled is None:
                    a1, ak1, a2, ak2, pro, psc, psh = src
                    _gemm(lib, dev, a1, ak1, a2, ak2, pro, None if pro == PRO_NONE else (psc, psh, None, None, None, None),
                          None, 0, w2, w2.stride(0), 0, b, z, M, Co, stat)
                    fuse_out = fuse_out and pool_k == 0
                scale, shift = torch.empty(Co, **f32), torch.empty(Co, **f32)
                bn = bns[l]
                out_args = None
                if fuse_out:
                    rows_out = M // 32 if pooled is not None else M
                    y = torch.empty((rows_out, Co), **f32)
                    argk = torch.empty((rows_out, Co), dtype=torch.uint8, device=dev) if pooled is not None else None
                    pm = pooled or (None, None, None, None)
                    out_args = (_ptr(z), Co if z is None else z.stride(0), _ptr(pm[0]), _ptr(pm[1]), _ptr(pm[2]), _ptr(pm[3]),
                                rows_out, _ptr(y), _ptr(argk), _stream(dev))
                if training:
                    if l == 0:
                        invalidate_eval_coefficients()     # running statistics are about to be rewritten by raw kernels
                    mean, invstd = torch.empty(Co, **f32), torch.empty(Co, **f32)
                    mom, mom_dev = momentum_word(bn, dev)
                    track = bn.track_running_stats and bn.running_mean is not None
                    if mom < 0.0 and track and bn.num_batches_tracked is not None:
                        bn.num_batches_tracked.add_(1)        # momentum=None: the kernel reads the bumped counter
                    fin_args = (_ptr(stat), P, Co, float(M), _ptr(gamma), _ptr(beta), float(bn.eps), mom,
                                _ptr(mom_dev), _ptr(bn.running_mean) if track else None,
                                _ptr(bn.running_var) if track else None, _ptr(scale), _ptr(shift),
                                _ptr(mean), _ptr(invstd),
                                _ptr(bn.num_batches_tracked) if track and bn.num_batches_tracked is not None else None)
                    if out_args is not None:
                        rc = lib.pn2_bn_finalize_out(*fin_args, *out_args)
                        _lib.check(rc, "pn2_bn_finalize_out")
                    else:
                        rc = lib.pn2_bn_finalize(*fin_args, _stream(dev))
                        _lib.check(rc, "pn2_bn_finalize")
                    coefs.append((scale, shift, mean, invstd))
                else:
                    scale, shift = _eval_coefficients(lib, dev, bn, gamma, beta)
                    if out_args is not None:
                        rc = lib.pn2_bn_finalize_out(None, 0, Co, 1.0, None, None, float(bn.eps), 0.0, None, None, None,
                                                     _ptr(scale), _ptr(shift), None, None, None, *out_args)
                        _lib.check(rc, "pn2_bn_finalize_out")
                    # mean / invstd of the frozen statistics: only needed if someone back-propagates
                    # through an eval-mode stack (BatchNorm is then a fixed affine map)
                    coefs.append((scale, shift, bn.running_mean.detach().clone(),
                                  torch.rsqrt(bn.running_var.detach() + bn.eps)))
                zs.append(z)
            Co = params[4 * (L - 1)].shape[0]
            argk2, k2 = None, 0
            if y is not None:
                pass                                    # emitted by pn2_bn_finalize_out
            elif pool_k > 255:
                # the winning row of a group is recorded in 8 bits: pool in two stages, k1 <= 255 rows per sub-group
                # (with the BatchNorm + ReLU), then the k2 = pool_k / k1 sub-group maxima of each group (already
                # activated: identity coefficients) -- group_all pools a whole cloud (pointnet2_utils.py:141-158, :200)
                k1 = max(d for d in range(1, 256) if pool_k % d == 0)
                k2 = pool_k // k1
                if k1 == 1 or k2 > 255:
                    raise NotImplementedError("max-pool over %d rows: no two-stage split with both factors <= 255" % pool_k)
                y1 = torch.empty((M // k1, Co), **f32)
                argk = torch.empty((M // k1, Co), dtype=torch.uint8, device=dev)
                rc = lib.pn2_bn_relu_out(_ptr(zs[-1]), M // k1, Co, k1, _ptr(coefs[-1][0]), _ptr(coefs[-1][1]), _ptr(y1),
                                         _ptr(argk), _stream(dev))
                _lib.check(rc, "pn2_bn_relu_out")
                y = torch.empty((M // pool_k, Co), **f32)
                argk2 = torch.empty((M // pool_k, Co), dtype=torch.uint8, device=dev)
                one, zero = torch.ones(Co, **f32), torch.zeros(Co, **f32)
                rc = lib.pn2_bn_relu_out(_ptr(y1), M // pool_k, Co, k2, _ptr(one), _ptr(zero), _ptr(y), _ptr(argk2), _stream(dev))
                _lib.check(rc, "pn2_bn_relu_out")
                pool_k = k1
            else:
                if pool_k:
                    rows_out = M // pool_k
                    y = torch.empty((rows_out, Co), **f32)
                    argk = torch.empty((rows_out, Co), dtype=torch.uint8, device=dev)
                else:
                    rows_out = M
                    y = torch.empty((M, Co), **f32)
                    argk = None
                rc = lib.pn2_bn_relu_out(_ptr(zs[-1]), rows_out, Co, pool_k, _ptr(coefs[-1][0]), _ptr(coefs[-1][1]), _ptr(y),
                                         _ptr(argk), _stream(dev))
                _lib.check(rc, "pn2_bn_relu_out")
        if _GATE_TAPS is not None:
            _GATE_TAPS.append({"zs": zs, "coefs": coefs, "argk": argk, "pool_k": pool_k, "y": y})
        ctx.training = training
        ctx.pool_k = pool_k
        ctx.argk2, ctx.k2 = argk2, k2
        ctx.L = L
        ctx.has_x2 = x2 is not None
        ctx.argk = argk
        ctx.coefs = coefs
        ctx.zs = zs
        ctx.wpad = wpad
        ctx.save_for_backward(x1, *((x2,) if x2 is not None else ()), *params)
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        x1 = saved[0]
        x2 = saved[1] if ctx.has_x2 else None
        params = saved[2 if ctx.has_x2 else 1:]
        L, pool_k, zs, coefs, argk = ctx.L, ctx.pool_k, ctx.zs, ctx.coefs, ctx.argk
        dev = gy.device
        lib = _lib.load()
        M, K1 = x1.shape
        K2 = 0 if x2 is None else x2.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        gy = gy.to(torch.float32).contiguous()
        if ctx.k2:
            # second pooling stage: route the pooled gradient to the winning sub-group of each group (small tensors)
            g1 = torch.zeros((gy.shape[0], ctx.k2, gy.shape[1]), **f32)
            g1.scatter_(1, ctx.argk2.to(torch.int64).unsqueeze(1), gy.unsqueeze(1))
            gy = g1.reshape(gy.shape[0] * ctx.k2, gy.shape[1])
        grads = [None] * (4 * L)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if (_DW_SIDE or M <= _DW_SIDE_MAX_ROWS) else None
        with torch.cuda.device(dev):
            # BatchNorm+ReLU backward statistics of the top layer
            zt = zs[-1]
            Ct = zt.shape[1]
            sc, sh, mu, istd = coefs[-1]
            rows = gy.shape[0]
            P = lib.pn2_bn_bwd_reduce_partials(rows)
            part = torch.empty((P, 2, Ct), **f32)
            rc = lib.pn2_bn_bwd_reduce(_ptr(gy), gy.stride(0), _ptr(zt), zt.stride(0), rows, Ct, _ptr(argk), pool_k,
                                       _ptr(sc), _ptr(sh), _ptr(mu), _ptr(istd), _ptr(part), _stream(dev))
            _lib.check(rc, "pn2_bn_bwd_reduce")
            g = gy
            g_argk = argk
            carried = None
            for l in range(L - 1, -1, -1):
                w, b = params[4 * l], params[4 * l + 1]
                Co = w.shape[0]
                w2 = w.reshape(Co, -1)
                cin = w2.shape[1]                       # columns of the gradient; Ci = columns of the rows
                if l == 0 and ctx.wpad is not None:
                    w2 = ctx.wpad
                Ci = w2.shape[1]
                z = zs[l]
                sc, sh, mu, istd = coefs[l]
                if carried is not None:                 # finalized in the launch that reduced the layer above's dW
                    dgamma, dbeta, c1, c2 = carried
                    carried = None
                else:
                    dgamma, dbeta = torch.empty(Co, **f32), torch.empty(Co, **f32)
                    c1, c2 = torch.empty(Co, **f32), torch.empty(Co, **f32)
                    rc = lib.pn2_bn_bwd_finalize(_ptr(part), P, Co, float(M), _ptr(dgamma), _ptr(dbeta), _ptr(c1), _ptr(c2),
                                                 _stream(dev))
                    _lib.check(rc, "pn2_bn_bwd_finalize")
                if not ctx.training:                    # frozen statistics: dz = scale * gh, no batch terms
                    c1.zero_()
                    c2.zero_()
                grads[4 * l + 2], grads[4 * l + 3] = dgamma, dbeta
                consts = (sc, sh, mu, istd, c1, c2)
                Pf = 0
                if _FUSED_BWD and (l > 0 or x2 is None):
                    Pf = lib.pn2_mlp_bwd_layer_partials(M, Co, Ci)
                if Pf:
                    pk = pool_k if g_argk is not None else 0
                    wpart = torch.empty((Pf, Co, Ci + 1), **f32)
                    dw, db = torch.empty((Co, cin), **f32), torch.empty(Co, **f32)
                    if l > 0:
                        xin, below = zs[l - 1], coefs[l - 1]
                        gp = torch.empty((M, Ci), **f32)
                        spart = torch.empty((Pf, 2, Ci), **f32)
                        nxt = tuple(torch.empty(Ci, **f32) for _ in range(4))    # dgamma, dbeta, c1, c2 of layer l-1
                    else:
                        xin, below = x1, (None, None, None, None)
                        gp = torch.empty((M, Ci), **f32) if ctx.needs_input_grad[2] else None
                        spart = None
                        nxt = (None, None, None, None)
                    rc = lib.pn2_mlp_bwd_layer(_ptr(g), g.stride(0), _ptr(z), z.stride(0), _ptr(g_argk), pk, _ptr(sc), _ptr(sh),
                                               _ptr(mu), _ptr(istd), _ptr(c1), _ptr(c2), _ptr(w2), w2.stride(0), _ptr(xin),
                                               xin.stride(0), _ptr(below[0]), _ptr(below[1]), _ptr(below[2]), _ptr(below[3]),
                                               _ptr(gp), 0 if gp is None else gp.stride(0), _ptr(spart), _ptr(wpart),
                                               _ptr(dw) if l > 0 else None, _ptr(db), _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]),
                                               _ptr(nxt[3]), M, Co, Ci, _stream(dev))
                    _lib.check(rc, "pn2_mlp_bwd_layer")
                    if l == 0:                           # the bottom layer's slabs: summed now or at the end of backward
                        finish_slabs(wpart, Pf, Co, Ci, cin, dw, db)
                    grads[4 * l], grads[4 * l + 1] = dw.view(w.shape), db.view(-1)
                    if l > 0:
                        g, g_argk, part, P = gp, None, spart, Pf
                        carried = nxt
                    else:
                        gx1, gx2 = gp, None
                    continue
                # dW, db
                Pw = lib.pn2_mlp_dw_partials(M, Co, Ci)
                wpart = torch.empty((Pw, Co, Ci + 1), **f32)
                dw, db = torch.empty((Co, cin), **f32), torch.empty(Co, **f32)
                if l == 0:
                    a1, a2, ak1, ak2, asc, ash = x1, x2, K1, K2, None, None
                else:
                    a1, a2, ak1, ak2 = zs[l - 1], None, zs[l - 1].shape[1], 0
                    asc, ash = coefs[l - 1][0], coefs[l - 1][1]
                if side is not None:
                    side.wait_stream(main)              # c1/c2, g and z of this layer are ready on main
                    dw_stream = side.cuda_stream
                else:
                    dw_stream = _stream(dev)
                # with a layer below and no side stream, the slabs stay unreduced here and are summed in the launch
                # that also finalizes the statistics the dX GEMM is about to produce (pn2_mlp_bwd_post)
                defer = l > 0 and side is None
                late = l == 0                           # bottom layer: pn2_mlp_dw_reduce_many, now or at the end of backward
                rc = lib.pn2_mlp_dw(_ptr(g), g.stride(0), _ptr(z), z.stride(0), _ptr(g_argk), pool_k if g_argk is not None else 0,
                                    _ptr(sc), _ptr(sh), _ptr(mu), _ptr(istd), _ptr(c1), _ptr(c2), _ptr(a1), a1.stride(0), ak1,
                                    _ptr(a2), 0 if a2 is None else a2.stride(0), ak2, _ptr(asc), _ptr(ash), M, Co, _ptr(wpart),
                                    None if (defer or late) else _ptr(dw), _ptr(db), dw_stream)
                _lib.check(rc, "pn2_mlp_dw")
                if late and side is not None:
                    with torch.cuda.stream(side):
                        reduce_slabs([(wpart, Pw, Co, Ci, cin, dw, db)], dev)
                elif late:
                    finish_slabs(wpart, Pw, Co, Ci, cin, dw, db)
                grads[4 * l], grads[4 * l + 1] = dw.view(w.shape), db.view(-1)
                # dX (= gradient w.r.t. the activation below), masked + reduced for the layer below
                if l > 0:
                    zp = zs[l - 1]
                    psc, psh, pmu, pistd = coefs[l - 1]
                    gp = torch.empty((M, Ci), **f32)
                    P = lib.pn2_mlp_gemm_max_partials(M)
                    part = torch.empty((P, 2, Ci), **f32)
                    _gemm(lib, dev, g, Co, z, Co, PRO_BN_BWD, consts, g_argk, pool_k if g_argk is not None else 0, w2,
                          w2.stride(0), 1, None, gp, M, Ci, part, (zp, psc, psh, pmu, pistd))
                    if defer:
                        carried = tuple(torch.empty(Ci, **f32) for _ in range(4))
                        rc = lib.pn2_mlp_bwd_post(_ptr(wpart), Pw, Co, Ci, _ptr(dw), _ptr(db), _ptr(part), P, Ci, float(M),
                                                  _ptr(carried[0]), _ptr(carried[1]), _ptr(carried[2]), _ptr(carried[3]),
                                                  _stream(dev))
                        _lib.check(rc, "pn2_mlp_bwd_post")
                    g, g_argk = gp, None
                else:
                    gx1 = gx2 = None
                    need1 = ctx.needs_input_grad[2]
                    need2 = ctx.has_x2 and ctx.needs_input_grad[3]
                    if need1 or need2:
                        # the two halves of the input gradient land in two dense tensors (no slicing copies)
                        gx1 = torch.empty((M, K1), **f32)
                        gx2 = torch.empty((M, K2), **f32) if K2 else None
                        _gemm(lib, dev, g, Co, z, Co, PRO_BN_BWD, consts, g_argk, pool_k if g_argk is not None else 0, w2,
                              w2.stride(0), 1, None, gx1, M, Ci, out2=gx2, nsplit=K1 if K2 else 0)
                        gx1 = gx1 if need1 else None
                        gx2 = gx2 if need2 else None
            if side is not None:
                main.wait_stream(side)                  # join: every dW/db is complete before grads are used
        return (None, None, gx1, gx2) + tuple(grads)


def _pad_cols(w, kin):
    """w [Co, cin] -> [Co, kin] with zero columns behind cin (one launch)."""
    dev = _dev(w)
    lib = _lib.load()
    w = w.detach()
    Co, cin = w.shape
    out = torch.empty((Co, kin), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pn2_copy_pad_cols(_ptr(w), w.stride(0), cin, _ptr(out), kin, kin, Co, _stream(dev))
    _lib.check(rc, "pn2_copy_pad_cols")
    return out


def mlp_stack(x1, x2, convs, bns, pool_k=0):
    """Run [conv -> bn -> relu] x len(convs) on rows [x1 | x2] ([M,K1], [M,K2] or None); with
    pool_k > 0 the output is max-pooled over groups of pool_k consecutive rows."""
    _dev(x1, x2)
    x1 = x1.to(torch.float32)
    if x1.stride(-1) != 1:
        x1 = x1.contiguous()
    if x2 is not None:
        x2 = x2.to(torch.float32)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
    params = []
    for l, (conv, bn) in enumerate(zip(convs, bns)):
        if conv.bias is None or bn.weight is None or bn.bias is None:
            raise NotImplementedError("mlp_stack expects conv bias and affine BatchNorm (as the reference builds them)")
        w = conv.weight
        if l == 0:
            # rows may carry zero pad columns (16-byte aligned rows): the stack pads the weight to match and returns
            # the gradient in the weight's own shape
            cin = w.shape[1]
            kin = x1.shape[1] + (0 if x2 is None else x2.shape[1])
            if kin < cin or (kin > cin and x2 is not None):
                raise ValueError("input rows have %d columns, first conv expects %d" % (kin, cin))
        params += [w, conv.bias, bn.weight, bn.bias]
    return _MLPStack.apply(list(bns), pool_k, x1, x2, *params)
