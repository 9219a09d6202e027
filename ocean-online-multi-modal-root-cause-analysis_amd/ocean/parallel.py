"""Data-parallel execution of the online step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; this is the MI355X scaling path.  The
model shards over samples; three quantities couple the ranks and each is one
collective on torch.distributed's default backend ("nccl" == RCCL on ROCm,
"gloo" in CPU tests):

* gradients: one all-reduce of a single flat fp32 bucket per step, averaged;
* training-mode BatchNorm statistics of the dilated CNN: an all-reduce of each
  layer's fp64 [sum, sum of squares] block (forward) and [sum g, sum g*xhat]
  block (backward), so N ranks reproduce the single-process global-batch
  statistics;
* InfoNCE in-batch negatives: an all-gather of the positive projections whose
  backward is a reduce-scatter.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _AllGatherRows(torch.autograd.Function):
    """[B_loc, P] on every rank -> [world * B_loc, P]; backward: reduce-scatter(sum)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = torch.empty(world * x.shape[0], *x.shape[1:], device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size(ctx.group)
        g = g.contiguous()
        out = torch.empty(g.shape[0] // world, *g.shape[1:], device=g.device, dtype=g.dtype)
        if dist.get_backend(ctx.group) == "gloo":        # gloo has no reduce_scatter: all-reduce then slice
            dist.all_reduce(g, group=ctx.group)
            r = dist.get_rank(ctx.group)
            out.copy_(g[r * out.shape[0]:(r + 1) * out.shape[0]])
        else:
            dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    return _AllGatherRows.apply(x, group)


class GradBucket:
    """Flat fp32 bucket over the gradients that exist after backward: pack, one
    all-reduce, average, unpack.  The bucket is reused across steps."""

    def __init__(self, group=None):
        self.group = group
        self._flat: Optional[torch.Tensor] = None

    def all_reduce_mean(self, grads: List[torch.Tensor]) -> None:
        if not grads:
            return
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        views, off = [], 0
        for g in grads:
            v = self._flat[off: off + g.numel()].view_as(g)
            views.append(v)
            off += g.numel()
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self._flat, group=self.group)
        self._flat.div_(dist.get_world_size(self.group))
        torch._foreach_copy_(grads, views)


class DataParallel:
    """Wires a (model, learner) pair for N-rank data parallelism.

    After construction every rank holds identical parameters and buffers
    (broadcast from rank 0); learner.grad_sync averages gradients; the dilated
    CNN all-reduces its BatchNorm statistics (sync_bn) and the contrastive
    module gathers in-batch negatives from every rank."""

    def __init__(self, model, learner=None, group=None, sync_bn: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model, self.group = model, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, src=0, group=group)
        self.bucket = GradBucket(group)
        if sync_bn:
            enc = model.temporal_encoder
            enc.stats_all_reduce = lambda t: dist.all_reduce(t, group=group)
            enc.stats_world_size = self.world
        cm = model.fusion_module.contrastive_module
        cm.gather_positives = lambda z: all_gather_rows(z, group)
        cm.rank_row_offset = lambda local_rows: self.rank * local_rows
        if learner is not None:
            learner.grad_sync = self.sync_grads

    def sync_grads(self, model=None) -> None:
        model = model or self.model
        # every rank must contribute the same tensors: parameters without a gradient on this rank
        # (none on the OCEAN path) would desynchronise the bucket, so the set is taken by name order
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        self.bucket.all_reduce_mean(grads)
