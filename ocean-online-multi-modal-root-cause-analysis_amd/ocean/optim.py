"""Fused gradient clipping + Adam for the online update step.

Drop-in for the pair the reference uses at
ocean/models/training/online_learner.py:325-335 / 402-412:
``clip_grad_norm_(params, max_norm)`` followed by ``torch.optim.Adam.step()``
(Adam built at :256-261: lr, betas (0.9, 0.999), eps 1e-8, coupled L2
weight_decay).  State keys match torch.optim.Adam ('step', 'exp_avg',
'exp_avg_sq') so checkpoints interchange.  Parameters whose .grad is None
are skipped and keep no state, exactly like torch.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _hip

# Page-locked table buffers that took part in a stream capture are never handed back to PyTorch's host allocator: the
# allocator would later query the event it recorded for them on the capturing stream, which HIP refuses
# (hipErrorCapturedEvent) and which invalidates whatever capture is running at that moment.  They are small (40 bytes per
# 2048 parameters); buffers of dropped recordings are reused by later ones.
_CAPTURED_PINNED: list = []

_CHUNK = 2048     # elements per block of the update kernels: ~1700 blocks for 3.3 M parameters (330 blocks of 16384 left most CUs idle)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}                    # per parameter group: chunk table, its key, chunk ranges per step count
        self._table = None                   # the first group's table (what a recorded step keeps alive)
        self._n_chunks = 0
        self._sq = None                      # fp64[3] on the device: ocean_adam_tail (accumulator, last step's sum, block counter)
        self._sq_clean = True                # the accumulator is zero (the self-resetting step leaves it so)
        self.last_grad_norm: Optional[torch.Tensor] = None      # device scalar, set when max_norm is given
        # HIP-graph support: learning rate and step count mirrored in device memory (see make_capturable)
        self._lr_dev: Optional[torch.Tensor] = None
        self._step_dev: Optional[torch.Tensor] = None
        self._lr_host: Optional[float] = None
        self._table_host = None
        self._table_free: list = []          # pinned table buffers of dropped recordings (see _CAPTURED_PINNED)
        self.generation = 0                  # bumped when the state tensors are replaced (load_state_dict)

    def _build_table(self, entries, device):
        rows = []
        for p, g, m, v in entries:
            n, base = p.numel(), 0
            while base < n:
                c = min(_CHUNK, n - base)
                rows.append((p.data_ptr() + 4 * base, g.data_ptr() + 4 * base, m.data_ptr() + 4 * base, v.data_ptr() + 4 * base, c))
                base += c
        # pinned staging + asynchronous copy: legal inside a stream capture, where it becomes a copy node that
        # re-reads this host buffer on every replay (so the buffer is kept alive and unchanged while the graph lives)
        table = torch.tensor(rows, dtype=torch.int64)                            # 5 x int64 per row == ocean_adam_chunk
        host = None
        for i, buf in enumerate(self._table_free):
            if buf.shape == table.shape:
                host = self._table_free.pop(i)
                break
        if host is None:
            host = torch.empty(table.shape, dtype=torch.int64, pin_memory=True)
        if torch.cuda.is_current_stream_capturing() and all(host is not b for b in _CAPTURED_PINNED):
            _CAPTURED_PINNED.append(host)       # wherever it was allocated: the copy below marks it on the capturing stream
        host.copy_(table)
        self._table_host = host
        return host.to(device, non_blocking=True), len(rows)

    def recycle_table_host(self, host: Optional[torch.Tensor]) -> None:
        """Called when a recorded step that used `host` as its table source is dropped: the buffer may serve the next
        recording.  (A buffer that backs a current table is never recycled: a recorded step re-reads it on replay.)"""
        current = any(host is sl["host"] for sl in self._tables.values())
        if host is not None and not current and all(host is not b for b in self._table_free):
            self._table_free.append(host)

    def common_step(self, params=None):
        """The step count shared by `params` (None: every parameter with state), or None when they differ.  Parameters
        without state yet count as 0."""
        if params is None:
            steps = {int(st["step"]) for st in self.state.values() if st}
        else:
            steps = {int(self.state[p]["step"]) if self.state.get(p) else 0 for p in params}
        if len(steps) > 1:
            return None
        return steps.pop() if steps else 0

    def make_capturable(self, device, params=None) -> None:
        """Keep the learning rate and the step count in device memory so that step() can be recorded into a HIP
        graph (torch.cuda.graph) and replayed: the kernels then read them instead of launch-time constants.
        The parameters the step will update (`params`; None: all with state) must share one step count: they do
        unless some were frozen for a while and then released."""
        k = self.common_step(params)
        if k is None:
            raise _hip.OceanHipError("capturable FusedAdam needs one common step count")
        steps = {k}
        self._step_dev = torch.full((1,), steps.pop() if steps else 0, dtype=torch.int32, device=device)
        self._lr_host = float(self.param_groups[0]["lr"])
        self._lr_dev = torch.full((1,), self._lr_host, dtype=torch.float32, device=device)

    def after_replay(self, params=None) -> None:
        """Host bookkeeping after one graph replay of step(): advance the step counters kept for state_dict() of the
        parameters the recorded step updates (`params`; None: every parameter with state) and push a changed learning
        rate to the device.  Parameters that were frozen when the step was recorded keep their count, as under
        torch.optim.Adam, which skips parameters without a gradient."""
        self.bump_steps(params, +1)
        self.sync_lr()

    def bump_steps(self, params, delta: int) -> None:
        states = self.state.values() if params is None else (self.state.get(p) for p in params)
        for st in states:
            if st:
                st["step"] = int(st["step"]) + delta

    def state_dict(self):
        """torch.optim.Adam's layout: 'step' leaves as a float tensor, as Adam writes it."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if "step" in st and not torch.is_tensor(st["step"]):
                st["step"] = torch.tensor(float(st["step"]))
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if st and torch.is_tensor(st.get("step")):
                st["step"] = int(st["step"])
        # the moment tensors are new objects and the step counts may differ: the chunk table, the device-side
        # mirrors and every recorded step that points at the old tensors are stale from here on
        self._tables = {}
        self._step_dev = self._lr_dev = self._lr_host = None
        self.generation += 1

    def sync_lr(self) -> None:
        if self._lr_dev is not None:
            lr = float(self.param_groups[0]["lr"])
            if lr != self._lr_host:
                self._lr_dev.fill_(lr)
                self._lr_host = lr

    def settle(self) -> None:
        """Before a step is recorded: put the norm accumulator into the state the self-resetting step expects, now, so
        that the recording does not contain a fill for it."""
        if self._sq is not None and not self._sq_clean:
            self._sq.zero_()
            self._sq_clean = True

    def reserve_table_host(self, params) -> None:
        """Page-lock a chunk-table buffer sized for `params` now, so that a step() recorded into a HIP graph right
        after finds it in the free list instead of calling the pinned allocator while the stream is capturing."""
        rows = sum((p.numel() + _CHUNK - 1) // _CHUNK for p in params)
        if rows and not any(b.shape[0] == rows for b in self._table_free):
            self._table_free.append(torch.empty((rows, 5), dtype=torch.int64, pin_memory=True))

    def _group_entries(self, group):
        """(step count -> [(param, grad, exp_avg, exp_avg_sq)]) of one parameter group, state created on first use and
        the per-parameter step counts advanced, as torch.optim.Adam.step does."""
        by_step = {}
        for p in group["params"]:
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise _hip.OceanHipError("FusedAdam needs contiguous fp32 parameters and gradients on the GPU")
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            # the count is a Python int while training (a tensor only right after loading a torch.optim.Adam
            # checkpoint): 126 CPU-tensor increments and reads per step were a measurable part of the host time
            k = int(st["step"]) + 1
            st["step"] = k
            by_step.setdefault(k, []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
        return by_step

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0, write_clipped: bool = True):
        """One Adam update.  max_norm > 0 first scales every gradient by
        min(1, max_norm / (global_norm + 1e-6)) like clip_grad_norm_: the norm is taken over the gradients of ALL
        parameter groups, as clip_grad_norm_(model.parameters()) does."""
        loss = closure() if closure is not None else None
        L = _hip.lib()
        if self._lr_dev is not None and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        plans = []
        for gi, group in enumerate(self.param_groups):
            by_step = self._group_entries(group)
            if not by_step:
                continue
            device = next(iter(by_step.values()))[0][0].device
            all_entries = [e for k in sorted(by_step) for e in by_step[k]]
            # the table holds addresses only: it is rebuilt when a tensor moved or the parameters fall into other
            # step-count groups, not because the counts themselves advanced (they do on every step)
            key = tuple((e[0].data_ptr(), e[1].data_ptr(), e[2].data_ptr()) for e in all_entries) + tuple(
                len(v) for _, v in sorted(by_step.items()))
            slot = self._tables.get(gi)
            if slot is None or slot["key"] != key:
                table, n_chunks = self._build_table(all_entries, device)
                # chunk counts per distinct step count (normally one)
                counts = [sum((e[0].numel() + _CHUNK - 1) // _CHUNK for e in by_step[k]) for k in sorted(by_step)]
                slot = self._tables[gi] = {"key": key, "table": table, "n": n_chunks, "counts": counts, "host": self._table_host}
            ranges, pos = [], 0
            for k, n in zip(sorted(by_step), slot["counts"]):
                ranges.append((pos, n, k))
                pos += n
            slot["ranges"] = ranges
            plans.append((group, slot, device))
        if not plans:
            return loss
        self._table, self._n_chunks, self._table_host = plans[0][1]["table"], plans[0][1]["n"], plans[0][1]["host"]
        sq_ptr = ctypes.c_void_p(0)
        device = plans[0][2]
        if self._sq is None or self._sq.device != device:
            self._sq = torch.zeros(3, dtype=torch.float64, device=device)
            self._sq_clean = True
        # device-side learning rate and step count, one group, one step count: norm + update need nothing in between
        # (no fill of the accumulator, no increment of the count - the update kernel's last block does both)
        if (self._step_dev is not None and len(plans) == 1 and len(self.param_groups) == 1 and len(plans[0][1]["ranges"]) == 1):
            group, slot, _ = plans[0]
            b1, b2 = group["betas"]
            if not self._sq_clean:
                self._sq.zero_()
                self._sq_clean = True
            _hip.check(L.ocean_clip_adam_step_auto(ctypes.c_void_p(slot["table"].data_ptr()), slot["n"], ctypes.c_void_p(self._sq.data_ptr()),
                                                   float(max_norm or 0.0), ctypes.c_void_p(self._lr_dev.data_ptr()),
                                                   ctypes.c_void_p(self._step_dev.data_ptr()), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]), int(write_clipped),
                                                   _hip.stream()), "ocean_clip_adam_step_auto")
            if max_norm and max_norm > 0:
                self.last_grad_norm = self._sq[1:2]
            return loss
        if max_norm and max_norm > 0:
            self._sq[0:1].zero_()
            self._sq_clean = False
            sq_ptr = ctypes.c_void_p(self._sq.data_ptr())
            for _, slot, _ in plans:            # one sum of squares over every group, then every group clips by it
                _hip.check(L.ocean_grad_sqnorm(ctypes.c_void_p(slot["table"].data_ptr()), slot["n"], sq_ptr, _hip.stream()),
                           "ocean_grad_sqnorm")
            self.last_grad_norm = self._sq[0:1]
        for group, slot, _ in plans:
            b1, b2 = group["betas"]
            tptr = slot["table"].data_ptr()
            if self._step_dev is not None and (len(slot["ranges"]) != 1 or len(self.param_groups) != 1):
                if torch.cuda.is_current_stream_capturing():
                    raise _hip.OceanHipError("capturable FusedAdam needs one parameter group and one common step count")
                # e.g. a component released after a freeze: its parameters are behind.  The device-side step count
                # cannot serve two counts: go back to launch-time constants and invalidate recorded steps.
                self._step_dev = self._lr_dev = self._lr_host = None
                self.generation += 1
            if self._step_dev is not None:
                self._step_dev.add_(1)
                _hip.check(L.ocean_clip_adam_step_dev(ctypes.c_void_p(tptr), slot["n"], sq_ptr,
                                                      float(max_norm or 0.0), ctypes.c_void_p(self._lr_dev.data_ptr()),
                                                      ctypes.c_void_p(self._step_dev.data_ptr()), float(b1), float(b2),
                                                      float(group["eps"]), float(group["weight_decay"]), int(write_clipped),
                                                      _hip.stream()), "ocean_clip_adam_step_dev")
                continue
            for pos, n, k in slot["ranges"]:
                _hip.check(L.ocean_clip_adam_step(ctypes.c_void_p(tptr + 40 * pos), n, sq_ptr, float(max_norm or 0.0),
                                                  float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                  float(group["weight_decay"]), int(k), int(write_clipped), _hip.stream()),
                           "ocean_clip_adam_step")
        return loss

    def grad_norm(self) -> Optional[torch.Tensor]:
        """sqrt of the squared norm accumulated by the last clipped step (device tensor)."""
        return None if self.last_grad_norm is None else self.last_grad_norm.sqrt().float()
