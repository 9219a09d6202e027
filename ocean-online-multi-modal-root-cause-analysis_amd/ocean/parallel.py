"""Data-parallel execution of the online step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; this is the MI355X scaling path.  The
model shards over samples; these quantities couple the ranks, each through one
collective of torch.distributed's backend ("nccl" == RCCL on ROCm; "gloo" in
CPU tests and in one-GPU rehearsals):

* gradients: every p.grad is a view of one flat arena (ocean._hip.GradArena),
  which is all-reduced in place (average) - the attention / fusion / head part
  as soon as backward reaches the encoders, the encoders' part at the end;
* training-mode BatchNorm statistics of the dilated CNN: an all-reduce of each
  layer's fp64 [sum, sum of squares] block (forward) and [sum g, sum g*xhat]
  block (backward), so N ranks reproduce the single-process global-batch
  statistics;
* InfoNCE in-batch negatives: an all-gather of the projections whose backward
  is a reduce-scatter; memory-bank negatives: an all-gather of the rows every
  rank then writes into its (identical) copy of the ring;
* the Trainer's feature-diversity term: an all-gather of the fused rows.

Ranks may hold different numbers of rows in one step (sharded streams): the
learner tells DataParallel the local and the global row count of the step and
the losses and BatchNorm counts are weighted accordingly; a rank without data
takes part with weight 0.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _hip


# ---------------------------------------------------------------------------------------------- transports
class Transport:
    """The collectives the data-parallel step needs, on device tensors through torch.distributed (RCCL when the
    group's backend is "nccl").  `capturable`: the calls may be recorded into a HIP graph."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.capturable = self.backend == "nccl"

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(t, src=src, group=self.group)

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        """In place; op in {'sum', 'avg', 'max'}.  Returns a handle with wait() when async_op."""
        if op == "avg" and self.backend != "nccl":            # gloo has no averaging reduction
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                return _ThenScale(w, t, 1.0 / self.world)
            t.div_(self.world)
            return None
        rop = {"sum": dist.ReduceOp.SUM, "avg": dist.ReduceOp.AVG, "max": dist.ReduceOp.MAX}[op]
        return dist.all_reduce(t, op=rop, group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, x: torch.Tensor) -> None:
        """out (contiguous, world x the size of x) = the ranks' x one after the other, in rank order."""
        dist.all_gather_into_tensor(out.view(-1), x.contiguous().view(-1), group=self.group)

    def reduce_scatter(self, out: torch.Tensor, x: torch.Tensor) -> None:
        if self.backend == "gloo":                            # gloo has no reduce_scatter: all-reduce, then slice
            x = x.contiguous().clone()
            dist.all_reduce(x, group=self.group)
            out.copy_(x[self.rank * out.shape[0]:(self.rank + 1) * out.shape[0]])
        else:
            dist.reduce_scatter_tensor(out, x.contiguous(), group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)

    def abort(self) -> None:
        """End this rank's part in the group now: collectives its peers have pending fail instead of blocking.  For a rank
        that cannot complete a collective step (an error between two of its collectives)."""
        pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
        aborter = getattr(dist.distributed_c10d, "_abort_process_group", None)
        if self.backend == "nccl" and aborter is not None:
            aborter(pg)
        else:
            pg.shutdown() if hasattr(pg, "shutdown") else dist.destroy_process_group(pg)

    def quiesce(self) -> None:
        """Before a step with collectives is recorded into a HIP graph: finish the device's work, so that every
        collective of the eager steps has completed before the recording starts.  (The process group's watchdog thread
        goes on polling the events of collectives it still lists; ocean._hip.capture records in thread-local capture
        mode, in which the runtime refuses unsafe calls of the recording thread only, so those polls are legal.)"""
        if self.backend == "nccl":
            torch.cuda.synchronize()


class _ThenScale:
    def __init__(self, work, t, f):
        self.work, self.t, self.f = work, t, f

    def wait(self):
        self.work.wait()
        self.t.mul_(self.f)
        return True


class _Done:
    def wait(self):
        return True


class HostStagedTransport(Transport):
    """The same collectives with every message staged through host memory and sent over gloo: for ranks that SHARE
    one GPU (two RCCL ranks cannot), i.e. rehearsals and tests of the N > 1 control flow on a one-GPU box.  Every call
    synchronises with the device; not capturable; its speed means nothing."""

    def __init__(self, group=None):
        super().__init__(group)
        self.capturable = False

    def broadcast(self, t, src=0):
        c = t.detach().cpu()
        dist.broadcast(c, src=src, group=self.group)
        t.detach().copy_(c)

    def all_reduce(self, t, op="sum", async_op=False):
        c = t.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
        if op == "avg":
            c /= self.world
        t.detach().copy_(c)
        return _Done() if async_op else None

    def all_gather(self, out, x):
        c = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(c, x.detach().cpu().contiguous().view(-1), group=self.group)
        out.copy_(c.view(out.shape))

    def reduce_scatter(self, out, x):
        c = x.detach().cpu().contiguous()
        dist.all_reduce(c, group=self.group)
        out.copy_(c[self.rank * out.shape[0]:(self.rank + 1) * out.shape[0]])


# ---------------------------------------------------------------------------------------------- autograd collectives
class _AllGatherRows(torch.autograd.Function):
    """[B_loc, ...] on every rank -> [world * B_loc, ...]; backward: reduce-scatter(sum)."""

    @staticmethod
    def forward(ctx, x, transport):
        ctx.transport = transport
        out = torch.empty(transport.world * x.shape[0], *x.shape[1:], device=x.device, dtype=x.dtype)
        transport.all_gather(out, x)
        return out

    @staticmethod
    def backward(ctx, g):
        t = ctx.transport
        out = torch.empty(g.shape[0] // t.world, *g.shape[1:], device=g.device, dtype=g.dtype)
        t.reduce_scatter(out, g.contiguous())
        return out, None


def all_gather_rows(x: torch.Tensor, group_or_transport=None) -> torch.Tensor:
    t = group_or_transport if isinstance(group_or_transport, Transport) else Transport(group_or_transport)
    return _AllGatherRows.apply(x, t)


class _AllGatherRowsRagged(torch.autograd.Function):
    """counts[r] rows on rank r -> [sum(counts), ...] on every rank, in rank order; backward: every rank receives the sum
    over ranks of the gradients of its own rows.  The collectives move max(counts) row slots per rank (the tail of a
    shorter rank's slots is zero and is dropped on arrival).  A rank whose count is 0 may pass any number of rows (its
    stand-in batch): none of them is sent, and their gradient is zero."""

    @staticmethod
    def forward(ctx, x, transport, counts):
        counts = [int(c) for c in counts]
        n, slots = counts[transport.rank], max(max(counts), 1)
        if x.shape[0] < n:
            raise ValueError(f"rank {transport.rank} holds {x.shape[0]} rows, the step's table says {n}")
        send = torch.zeros(slots, *x.shape[1:], device=x.device, dtype=x.dtype)
        send[:n] = x[:n]
        out = torch.empty(transport.world * slots, *x.shape[1:], device=x.device, dtype=x.dtype)
        transport.all_gather(out, send)
        ctx.transport, ctx.counts, ctx.slots, ctx.local_rows = transport, counts, slots, x.shape[0]
        return torch.cat([out[r * slots: r * slots + c] for r, c in enumerate(counts)], dim=0)

    @staticmethod
    def backward(ctx, g):
        t, counts, slots = ctx.transport, ctx.counts, ctx.slots
        padded = torch.zeros(t.world * slots, *g.shape[1:], device=g.device, dtype=g.dtype)
        at = 0
        for r, c in enumerate(counts):
            padded[r * slots: r * slots + c] = g[at: at + c]
            at += c
        mine = torch.empty(slots, *g.shape[1:], device=g.device, dtype=g.dtype)
        t.reduce_scatter(mine, padded)
        gx = torch.zeros(ctx.local_rows, *g.shape[1:], device=g.device, dtype=g.dtype)
        n = counts[t.rank]
        gx[:n] = mine[:n]
        return gx, None, None


def all_gather_rows_ragged(x: torch.Tensor, counts, group_or_transport=None) -> torch.Tensor:
    t = group_or_transport if isinstance(group_or_transport, Transport) else Transport(group_or_transport)
    return _AllGatherRowsRagged.apply(x, t, counts)


class GradBucket:
    """Flat fp32 bucket over a list of gradient tensors: pack, one all-reduce, average, unpack.  Used for gradients that
    do not live in an arena (a model driven without OnlineLearner / Trainer); the bucket is reused across steps."""

    def __init__(self, group_or_transport=None):
        self.transport = group_or_transport if isinstance(group_or_transport, Transport) else None
        self._group = None if isinstance(group_or_transport, Transport) else group_or_transport
        self._flat: Optional[torch.Tensor] = None

    def all_reduce_mean(self, grads: List[torch.Tensor]) -> None:
        if not grads:
            return
        if self.transport is None:
            self.transport = Transport(self._group)
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, device=grads[0].device, dtype=torch.float32)
        views, off = [], 0
        for g in grads:
            views.append(self._flat[off: off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        self.transport.all_reduce(self._flat, "avg")
        torch._foreach_copy_(grads, views)


# ---------------------------------------------------------------------------------------------- the wiring
class DataParallel:
    """Wires a (model, learner) pair for N-rank data parallelism.

    After construction every rank holds identical parameters and buffers (broadcast from rank 0).  `learner` may be an
    OnlineLearner or a Trainer (both call learner.grad_sync(model) between backward and the optimizer step); its loss
    function / contrastive module / temporal encoder get the gather and statistics hooks.

    early_bucket (default on): the gradients of the attention / fusion / head stages are complete when backward reaches
    the temporal encoder, whose backward is most of the step; their slice of the arena is all-reduced from there,
    asynchronously, and the encoders' slice follows at the end.  It relies on every parameter being used once per
    backward pass (true of OCEANModel)."""

    def __init__(self, model, learner=None, group=None, sync_bn: bool = True, early_bucket: Optional[bool] = None,
                 transport: Optional[Transport] = None):
        self.transport = transport if transport is not None else Transport(group)
        self.model, self.group = model, group
        self.world, self.rank = self.transport.world, self.transport.rank
        self.capturable = self.transport.capturable
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                self.transport.broadcast(t, 0)
        self.bucket = GradBucket(self.transport)
        self.arena: Optional[_hip.GradArena] = None
        self._split = 0
        self._early_work = None
        if early_bucket is None:
            early_bucket = os.environ.get("OCEAN_DP_EARLY_BUCKET", "1") != "0"
        self.early_bucket = bool(early_bucket)
        # rows of this step: local, and summed over ranks (None: every rank holds the same number of rows)
        self._rows_local: Optional[int] = None
        self._rows_global: Optional[int] = None
        self._rows_all: Optional[List[int]] = None
        params = [p for p in model.parameters() if p.requires_grad]
        if params and params[0].is_cuda:
            # arena order: the two encoders first, everything after them last, so that either part is one slice
            enc = getattr(model, "temporal_encoder", None), getattr(model, "spatial_encoder", None)
            late_ids = {id(p) for e in enc if e is not None for p in e.parameters()}
            ordered = [p for p in params if id(p) in late_ids] + [p for p in params if id(p) not in late_ids]
            self.arena = _hip.GradArena(ordered)
            _hip.register_grad_arena(self.arena)
            n_late = sum((p.numel() + 3) // 4 * 4 for p in ordered if id(p) in late_ids)
            self._split = n_late if 0 < n_late < self.arena.flat.numel() else 0
            if self.early_bucket and self._split and getattr(model, "temporal_encoder", None) is not None:
                model.temporal_encoder.register_forward_hook(self._watch_encoder_output)
        enc = getattr(model, "temporal_encoder", None)
        if sync_bn and enc is not None:
            enc.stats_all_reduce = self._stats_all_reduce
            enc.stats_all_reduce_start = self._stats_all_reduce_start
            enc.stats_world_size = self.world
            enc.stats_global_rows = self._global_rows
        fm = getattr(model, "fusion_module", None)
        if fm is not None:
            cm = fm.contrastive_module
            cm.gather_positives = self._gather_positives
            cm.rank_row_offset = self._row_offset
            cm.gather_bank_rows = self._gather_bank_rows
        if learner is not None:
            learner.grad_sync = self.sync_grads
            learner.data_parallel = self
            lf = getattr(learner, "loss_function", None)
            if lf is not None:
                lf.gather_rows = lambda x: all_gather_rows(x, self.transport)

    # ------------------------------------------------------------------ ragged steps
    def set_step_rows(self, rows_per_rank: Optional[List[int]]) -> None:
        """Tell the wiring how many sample rows every rank holds in the coming step (sharded streams: the ranks'
        batches differ in size; a rank with none runs a one-row stand-in batch that counts as 0 rows).  None: equal
        batches on every rank (the default).  With unequal batches the gathers of projections (in-batch InfoNCE
        negatives, memory-bank rows) move max-count row slots per rank and keep each rank's true rows, so the negatives
        of every anchor are the rows of the whole step's batch, as in a single process (graph_fusion.py:144-157)."""
        if rows_per_rank is None:
            self._rows_local = self._rows_global = self._rows_all = None
        else:
            self._rows_all = [int(r) for r in rows_per_rank]
            self._rows_local, self._rows_global = self._rows_all[self.rank], sum(self._rows_all)

    def loss_weight(self) -> float:
        """Factor for this rank's (local-mean) loss so that the rank-averaged gradient is the gradient of the mean
        over ALL rows of the step: world * local / global (1 when batches are equal)."""
        if self._rows_global is None:
            return 1.0
        return self.world * float(self._rows_local) / float(max(self._rows_global, 1))

    def _global_rows(self, local_rows: int) -> int:
        return self.world * local_rows if self._rows_global is None else self._rows_global

    def _stats_all_reduce(self, t: torch.Tensor) -> None:
        if self._rows_local == 0:
            t.zero_()                      # a stand-in batch contributes nothing to the global statistics
        self.transport.all_reduce(t, "sum")

    def _stats_all_reduce_start(self, t: torch.Tensor):
        """The same reduction, started now and waited for later (backward pass: the weight gradient of the layer
        above does not depend on it and runs in between)."""
        if self._rows_local == 0:
            t.zero_()
        work = self.transport.all_reduce(t, "sum", async_op=True)
        return work if work is not None else _Done()

    # ------------------------------------------------------------------ in-batch negatives
    def _gather_positives(self, z: torch.Tensor) -> torch.Tensor:
        """[B_loc, W] projections of this rank -> the rows of all ranks in rank order, keeping autograd."""
        if self._rows_all is None:
            return all_gather_rows(z, self.transport)
        return all_gather_rows_ragged(z, self._rows_all, self.transport)

    def _row_offset(self, local_rows: int) -> int:
        """Where this rank's rows start in the gathered batch: the label of anchor i is offset + i.  (A stand-in
        batch has no rows there; its one anchor points at row 0 and its loss carries weight 0.)"""
        if self._rows_all is None:
            return self.rank * local_rows
        return sum(self._rows_all[:self.rank]) if self._rows_all[self.rank] > 0 else 0

    # ------------------------------------------------------------------ memory bank
    def _gather_bank_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """[M, B_loc, P] projections of this rank -> [M, B_all, P] of all ranks in rank order: the rows the
        single-process model would have written into the bank for the whole batch (graph_fusion.py:237-245), in its
        order.  No gradient (the bank is detached)."""
        M, B, P = rows.shape
        if self._rows_all is None:
            out = torch.empty(self.world, M, B, P, device=rows.device, dtype=rows.dtype)
            self.transport.all_gather(out.view(self.world * M, B, P), rows.detach().contiguous())
            return out.permute(1, 0, 2, 3).reshape(M, self.world * B, P)
        # unequal batches: every rank sends max-rows slots, the receivers keep each rank's true rows (a stand-in
        # batch has one row and counts as none)
        Bmax = max(max(self._rows_all), 1)
        send = torch.zeros(M, Bmax, P, device=rows.device, dtype=rows.dtype)
        send[:, :min(B, Bmax)] = rows.detach()[:, :Bmax]
        out = torch.empty(self.world, M, Bmax, P, device=rows.device, dtype=rows.dtype)
        self.transport.all_gather(out.view(self.world * M, Bmax, P), send)
        return torch.cat([out[r, :, :n] for r, n in enumerate(self._rows_all)], dim=1)

    # ------------------------------------------------------------------ gradients
    def _watch_encoder_output(self, module, inputs, output):
        if torch.is_tensor(output) and output.requires_grad and torch.is_grad_enabled():
            output.register_hook(self._start_early_slice)

    def _start_early_slice(self, grad):
        """Runs inside backward, when the gradient of the temporal encoder's output is known: every stage after the
        encoders has finished its backward, i.e. has written its parameter gradients into the arena."""
        if self._early_work is None and self.arena is not None and self._split:
            if self._rows_local == 0:
                self.arena.flat[self._split:].zero_()          # a stand-in batch contributes no gradient (see sync_grads)
            self._early_work = self.transport.all_reduce(self.arena.flat[self._split:], "avg", async_op=True)
        return None

    def reduce_losses(self, vec: torch.Tensor) -> torch.Tensor:
        """Local mean losses [k] -> the means over all rows of the step, identical on every rank (one small
        all-reduce).  The learner feeds these to its learning-rate policy, which must not diverge between ranks."""
        w = self.loss_weight()
        vec = vec * w if w != 1.0 else vec.clone()
        self.transport.all_reduce(vec, "avg")
        return vec

    def sync_grads(self, model=None) -> None:
        model = model or self.model
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        if not grads:
            return
        a = self.arena
        in_arena = False
        if a is not None:
            lo = a.flat.data_ptr()
            hi = lo + 4 * a.flat.numel()
            in_arena = all(lo <= g.data_ptr() < hi for g in grads)
        if not in_arena:
            # gradients that were not written into the arena (another driver produced them): bucket them
            if self._early_work is not None:
                self._early_work.wait()
                self._early_work = None
            self.bucket.all_reduce_mean(grads)
            return
        # A stand-in batch (no rows of its own this step) must add nothing.  Its loss weight is 0, but the synchronised
        # BatchNorm backward still hands its row the batch-mean terms of the OTHER ranks' gradients (its row took no
        # part in the statistics those terms belong to): its gradients are zero by definition, so they are zeroed.
        standin = self._rows_local == 0
        if self._early_work is not None:
            if standin:
                a.flat[:self._split].zero_()
            self.transport.all_reduce(a.flat[:self._split], "avg")
            self._early_work.wait()
            self._early_work = None
        else:
            if standin:
                a.flat.zero_()
            self.transport.all_reduce(a.flat, "avg")
