"""TEST HELPER (not product code): the gather / scatter form of the data-parallel gradient exchange -- all parameter gradients
copied into ONE flat fp32 buffer, one collective, copied back.  The product's trainers use fneus.parallel.GradArena (every
.grad a view of the arena, in-place all-reduce, no copies); tests/test_host_cpu.py checks the two forms against each other
under world-size-2 gloo."""
from typing import Iterable, List

import torch
import torch.distributed as dist

from fneus.parallel import _active


class FlatGradBucket:
    """All parameter gradients in ONE contiguous fp32 buffer (<= 8 MB on this path): a single collective per step."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off: off + p.numel()].view_as(p))
            off += p.numel()

    def gather(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def scatter(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                p.grad.copy_(v)

    def allreduce_sum(self, group=None):
        """plain sum over ranks: with global loss normalisers (reduce_loss_norms) every rank's gradient is already its
        share of the global batch's gradient"""
        if not _active(group):
            return
        self.gather()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.scatter()

    def allreduce_mean(self, group=None):
        """sum over ranks, divide by world size (every rank must call this every step: no data-dependent skipping)"""
        if not _active(group):
            return
        self.gather()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))
        self.scatter()
