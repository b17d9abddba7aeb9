"""Data-parallel training step for pointnet2_sem_seg: one process per GPU, replicas of the
968,914-parameter model, independent 4096-point blocks per rank, ONE all-reduce (RCCL over xGMI
when the backend is "nccl") of a flat fp32 gradient buffer per step.

The reference has no distributed code (single process, sem_seg_training.py:374); the step body
restates localfunctions.py:203-218: zero_grad -> forward -> nll_loss(weight) -> backward -> Adam
(sem_seg_training.py:576-582).  BatchNorm statistics stay rank-local (no SyncBN upstream)."""
import torch
import torch.distributed as dist


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
        if world == 1:
            return
        flat = self.pack()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)


class SemSegTrainer:
    def __init__(self, model, lr=1e-3, weight_decay=1e-4, class_weight=None, group=None):
        self.model = model
        self.group = group
        self.grads = FlatGradients(model)
        self.class_weight = class_weight
        fused = next(model.parameters()).is_cuda
        self.optimizer = torch.optim.Adam(self.grads.params, lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                          weight_decay=weight_decay, fused=fused)

    def broadcast_parameters(self, src=0):
        """Replicas start identical: rank `src`'s parameters and buffers go to every rank."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)

    def step(self, blocks_cf, target):
        """blocks_cf [B,C,N] (channel-first like the reference loop, localfunctions.py:209),
        target [B,N] int64.  Returns the (rank-local) loss tensor, no host sync."""
        self.model.train()
        self.grads.zero()
        pred, _ = self.model(blocks_cf)
        loss = torch.nn.functional.nll_loss(pred.reshape(-1, pred.shape[-1]), target.reshape(-1),
                                            weight=self.class_weight)
        loss.backward()
        self.grads.all_reduce_mean(self.group)
        self.optimizer.step()
        return loss.detach()
