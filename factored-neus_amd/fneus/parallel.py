"""Data-parallel ray sharding: one process per GPU, full replicas, one flat-bucket all-reduce per step.

The reference is single-GPU (exp_runner.py:651-661); rays are independent, so rank r renders its own B rays and the
only exchange is the gradient sum.  backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used
by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _single_rank_collectives() -> bool:
    """FNEUS_DP_SINGLE=1: issue every collective even in a world of ONE rank.  A one-GPU box can then run the exact RCCL
    call sequence of the data-parallel step (communicator set-up, the async all-reduce on the side stream, collectives
    between the hipGraph segments); the results are unchanged (a sum over one rank)."""
    return os.environ.get("FNEUS_DP_SINGLE", "0") == "1"


def collectives_active(group=None) -> bool:
    """are collectives issued?  (more than one rank, or FNEUS_DP_SINGLE=1)"""
    return _active(group)


def _active(group=None) -> bool:
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _single_rank_collectives())


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _single_rank_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                rank=rank, world_size=world)
    return rank, world, local


class GradArena:
    """ONE contiguous fp32 buffer holding the gradient of every parameter of the model.  The fused MLPs put their flat
    gradient buffers into slices of it, the small torch modules get persistent `.grad` views: the data-parallel exchange
    is a single in-place all-reduce of `flat` with no gather / scatter copies, and the optimiser sees adjacent segments."""

    def __init__(self, device, fused_modules, ref_color, small_modules, n_late: int = 0):
        """n_late: the first n_late fused modules form the LATE part of the arena (their gradients are complete only at
        the very end of the backward: the SDF network, the background NeRF); everything behind them -- colour network,
        RefColor heads, small modules -- is final when the SDF backward starts and can be exchanged beside it
        (allreduce_early / allreduce_late)."""
        sizes = [m.n_raw() for m in fused_modules] + (ref_color.n_raw() if ref_color is not None else [])
        small = [p for m in small_modules if m is not None for p in m.parameters()]
        total = sum(sizes) + sum(p.numel() for p in small)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        off = 0
        slices = []
        for n in sizes:
            slices.append(self.flat[off: off + n])
            off += n
        for m, sl in zip(fused_modules, slices):
            m.use_grad_buffer(sl)
        if ref_color is not None:
            ref_color.use_grad_buffers(slices[-2], slices[-1])
        for p in small:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        self.small = small
        self._small_views = [p.grad for p in small]
        split = sum(sizes[:n_late])
        self.late, self.early = self.flat[:split], self.flat[split:]

    def restore_small_grads(self):
        """something set a small parameter's .grad to None (or replaced it): point it back at its arena view"""
        for p, v in zip(self.small, self._small_views):
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def allreduce_sum(self, group=None):
        if _active(group):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    # ---- the exchange in two parts: `early` while the SDF backward still runs, `late` after it ----
    def allreduce_early(self, side_stream, group=None):
        """start the all-reduce of the early part on `side_stream`, ordered after everything issued so far on the current
        stream; the current stream does NOT wait.  -> a handle for wait_early()."""
        if not _active(group) or self.early.numel() == 0:
            return None
        if not self.flat.is_cuda or side_stream is None:      # host tensors (gloo tests): nothing to overlap with
            dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=group)
            return None
        side_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_stream):      # RCCL's stream waits for the stream that is current at the call
            work = dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=group, async_op=True)
        return work, side_stream

    def allreduce_late(self, group=None):
        if _active(group) and self.late.numel() > 0:
            dist.all_reduce(self.late, op=dist.ReduceOp.SUM, group=group)

    @staticmethod
    def wait_early(handle):
        """make the current stream wait for the early part's all-reduce"""
        if handle is None:
            return
        work, side_stream = handle
        with torch.cuda.stream(side_stream):
            work.wait()
        torch.cuda.current_stream().wait_stream(side_stream)


def reduce_loss_norms(norms: torch.Tensor, group=None) -> torch.Tensor:
    """[sum mask, sum mask*sdf_mask, sum eik_den, ray count] of this rank -> of the global batch (SURVEY.md section 8(e):
    the small all-reduce BEFORE the loss that makes R ranks x B rays equal to one R*B-ray batch).  In place."""
    if _active(group):
        dist.all_reduce(norms, op=dist.ReduceOp.SUM, group=group)
    return norms


def broadcast_parameters(modules, src: int = 0, group=None):
    if not _active(group):
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            # (detach(), not .data: the write then counts in the parameter's version, which tells a frozen network to pack again --
            #  models/fields.py _frozen_key)
            dist.broadcast(t.detach(), src=src, group=group)
