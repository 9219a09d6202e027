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

import os
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

    def start(self, grads: List[torch.Tensor]):
        """Pack and launch the all-reduce without waiting for it; finish() averages and unpacks.  The collective runs
        on the backend's own stream, ordered after the packing copies, so kernels queued afterwards overlap it."""
        self._views = self._pack(grads)
        self._grads = grads
        self._work = dist.all_reduce(self._flat, group=self.group, async_op=True) if grads else None

    def finish(self) -> None:
        if getattr(self, "_work", None) is not None:
            self._work.wait()                     # the current stream waits for the collective
            self._flat.div_(dist.get_world_size(self.group))
            torch._foreach_copy_(self._grads, self._views)
        self._work = self._grads = self._views = None

    def _pack(self, grads: List[torch.Tensor]) -> List[torch.Tensor]:
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        views, off = [], 0
        for g in grads:
            views.append(self._flat[off: off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        return views

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

    def __init__(self, model, learner=None, group=None, sync_bn: bool = True, early_bucket: Optional[bool] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model, self.group = model, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, src=0, group=group)
        self.bucket = GradBucket(group)
        # early_bucket (opt-in; OCEAN_DP_EARLY_BUCKET=1): gradients of the attention / fusion / head stages are complete
        # when backward reaches the temporal encoder, whose backward is most of the step; their all-reduce is started
        # there and runs under it, the encoders' gradients follow at the end.  Off by default: with one rank it costs
        # 0.15 ms of host time per step, and what it hides on N ranks has not been measured.
        self.early_bucket = GradBucket(group)
        self._early_ids = set()
        self._early_started = False
        if early_bucket is None:
            early_bucket = os.environ.get("OCEAN_DP_EARLY_BUCKET") == "1"
        if learner is not None and early_bucket:
            late = {id(p) for enc in (model.temporal_encoder, model.spatial_encoder) for p in enc.parameters()}
            self._early_ids = {id(p) for p in model.parameters() if id(p) not in late}
            model.temporal_encoder.register_forward_hook(self._watch_encoder_output)
        if sync_bn:
            enc = model.temporal_encoder
            enc.stats_all_reduce = lambda t: dist.all_reduce(t, group=group)
            enc.stats_world_size = self.world
        cm = model.fusion_module.contrastive_module
        cm.gather_positives = lambda z: all_gather_rows(z, group)
        cm.rank_row_offset = lambda local_rows: self.rank * local_rows
        if learner is not None:
            learner.grad_sync = self.sync_grads

    def _watch_encoder_output(self, module, inputs, output):
        if torch.is_tensor(output) and output.requires_grad and torch.is_grad_enabled():
            output.register_hook(self._start_early_bucket)

    def _start_early_bucket(self, grad):
        """Runs inside backward, when the gradient of the temporal encoder's output is known: every stage after the
        encoders has finished its backward (parameter gradients are accumulated before the engine moves on)."""
        if not self._early_started:
            grads = [p.grad for p in self.model.parameters() if id(p) in self._early_ids and p.grad is not None]
            if grads:
                self.early_bucket.start(grads)
                self._early_started = True
                self._early_set = {id(g) for g in grads}
        return None

    def sync_grads(self, model=None) -> None:
        model = model or self.model
        # every rank must contribute the same tensors: parameters without a gradient on this rank
        # (none on the OCEAN path) would desynchronise the bucket, so the set is taken by name order
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        if self._early_started:
            self.bucket.all_reduce_mean([g for g in grads if id(g) not in self._early_set])
            self.early_bucket.finish()
            self._early_started = False
        else:
            self.bucket.all_reduce_mean(grads)
