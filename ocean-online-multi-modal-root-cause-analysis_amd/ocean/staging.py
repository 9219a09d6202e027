"""Host -> HBM staging of one update batch.

The reference collates a batch with three torch.stack calls over per-sample CPU
tensors (ocean/models/training/online_learner.py:378-393, and the same in
ocean/data/datasets/multimodal_dataset.py:292-343) and lets the model's first
kernels pull the pageable result across PCIe synchronously.  Here the samples
are stacked straight into page-locked buffers that are allocated once per batch
shape and copied with asynchronous DMA into device tensors that are reused from
step to step, so a steady-state step allocates nothing on either side and the
copies run on a stream of their own.  A caller that knows its next batch (the
stream processor does: it cuts one polling cycle into several batches) can have it
collated while the GPU is still busy with the current step; the step then starts
on data that is already in HBM.

The values that reach the model are the same bytes either way.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch


class _Slot:
    def __init__(self, shapes: Sequence[Tuple[int, ...]], device: torch.device, pin: bool):
        self.host = [torch.empty(s, dtype=torch.float32, pin_memory=pin) for s in shapes]
        self.dev = [torch.empty(s, dtype=torch.float32, device=device) for s in shapes]
        cuda = device.type == "cuda"
        self.done = torch.cuda.Event() if cuda else None    # recorded on the compute stream after the slot's last reader
        self.ready = torch.cuda.Event() if cuda else None   # recorded on the copy stream after the slot's last DMA
        self.used = False


class BatchStager:
    """Ring of pinned host / device buffer sets, one ring per batch shape.

    collate(samples) returns (metrics [B,T,F], logs [B,...], labels [B,S]) on `device`.  The tensors returned by
    one call stay valid until `depth` further calls with the same shape; the update step consumes them before it
    returns, so the default depth of 2 leaves a full step of slack.
    """

    def __init__(self, device, depth: int = 2, max_shapes: int = 8):
        self.device = torch.device(device)
        self.depth, self.max_shapes = max(1, int(depth)), max_shapes
        self.pin = self.device.type == "cuda"
        self._rings: Dict[Tuple, List[_Slot]] = {}
        self._next: Dict[Tuple, int] = {}
        self._copy_stream = torch.cuda.Stream(self.device) if self.pin else None

    @staticmethod
    def stageable(samples) -> bool:
        """Cheap screen: the first and last sample carry fp32 CPU tensors of one shape (what the stream front-end
        produces).  A ragged batch that passes it makes collate() return None instead."""
        s0, s1 = samples[0], samples[-1]
        for a, b in ((s1.metrics, s0.metrics), (s1.logs, s0.logs), (s1.label, s0.label)):
            if (a.device.type != "cpu" or b.device.type != "cpu" or a.dtype != torch.float32 or b.dtype != torch.float32
                    or a.shape != b.shape):
                return False
        return True

    def _slot(self, key) -> _Slot:
        ring = self._rings.get(key)
        if ring is None:
            if len(self._rings) >= self.max_shapes:        # bounded page-locked memory: drop the oldest shape
                old = next(iter(self._rings))
                del self._rings[old], self._next[old]
            ring = self._rings[key] = [_Slot(key, self.device, self.pin) for _ in range(self.depth)]
            self._next[key] = 0
        i = self._next[key]
        self._next[key] = (i + 1) % self.depth
        return ring[i]

    def collate(self, samples) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """Device tensors of the stacked batch, or None when the samples do not stack into fp32 buffers of the
        first sample's shape (the caller then collates the reference's way, which raises the reference's errors)."""
        B = len(samples)
        s0 = samples[0]
        key = ((B, *s0.metrics.shape), (B, *s0.logs.shape), (B, *s0.label.shape))
        slot = self._slot(key)
        if slot.done is not None and slot.used:
            slot.done.synchronize()          # the step that read this slot's device tensors has finished
        try:
            for h, d, name in zip(slot.host, slot.dev, ("metrics", "logs", "label")):
                parts = [getattr(s, name) for s in samples]
                if parts[B // 2].dtype != torch.float32:
                    return None
                torch.stack(parts, out=h)             # raises on a shape / device mismatch inside the batch
                if self._copy_stream is None:
                    d.copy_(h)
                else:
                    with torch.cuda.stream(self._copy_stream):
                        d.copy_(h, non_blocking=True)     # DMA of this tensor overlaps the stacking of the next
                slot.used = True
        except RuntimeError:
            return None
        if slot.ready is not None:
            slot.ready.record(self._copy_stream)
        return slot.dev[0], slot.dev[1], slot.dev[2]

    def acquire(self, tensors) -> None:
        """Order the current stream after the DMA that filled collate()'s tensors; call before the first kernel
        that reads them."""
        slot = self._find(tensors)
        if slot is not None and slot.ready is not None:
            torch.cuda.current_stream(self.device).wait_event(slot.ready)

    def _find(self, tensors) -> Optional[_Slot]:
        for ring in self._rings.values():
            for slot in ring:
                if slot.dev[0] is tensors[0]:
                    return slot
        return None

    def release(self, tensors) -> None:
        """Mark the end of the step that consumed collate()'s tensors (records the slot's event on the current stream)."""
        slot = self._find(tensors)
        if slot is not None and slot.done is not None:
            slot.done.record(torch.cuda.current_stream(self.device))


class ReplayPool:
    """Device-resident copy of the samples a MemoryBuffer holds, for experience replay.

    The reference rebuilds every replay batch from the Python deque: three torch.stack calls over per-sample host
    tensors and a host-to-device transfer (ocean/models/training/online_learner.py:433-453 via :378-393), although
    each of those samples already crossed PCIe once when it was first processed.  Here a sample's three tensors are
    written into one row of three preallocated device tensors when the buffer takes it in, the row is handed back
    when the buffer evicts it, and a replay batch is three row gathers on the device.  Which samples make a batch
    is still decided by MemoryBuffer.sample (same random draws); the pool only changes where their bytes come from.

    The first sample fixes the row shapes; a sample of another shape or dtype, or one that arrives when no row is
    free, is simply not pooled, and gather() answers None for a batch that contains such a sample (the caller then
    collates that batch from the host copies).
    """

    def __init__(self, device, capacity: int):
        self.device, self.capacity = torch.device(device), int(capacity)
        self._rows: Optional[List[torch.Tensor]] = None          # [capacity, ...] per field
        self._shapes: Optional[Tuple] = None
        self._slot_of: Dict[int, int] = {}                       # id(sample) -> row
        self._owner: List[object] = [None] * self.capacity       # row -> the sample itself: keeps its id from being reused
        self._free: List[int] = list(range(self.capacity - 1, -1, -1))

    def __len__(self) -> int:
        return len(self._slot_of)

    def __contains__(self, sample) -> bool:
        return id(sample) in self._slot_of

    @staticmethod
    def _fields(sample):
        return sample.metrics, sample.logs, sample.label

    def put(self, sample) -> bool:
        """Copy the sample's tensors into a free row (asynchronously on the current stream when they sit in pinned
        memory, as an ordinary copy otherwise).  False when the sample cannot be pooled."""
        fields = self._fields(sample)
        if any((not torch.is_tensor(f)) or f.dtype != torch.float32 for f in fields):
            return False
        shapes = tuple(tuple(f.shape) for f in fields)
        if self._rows is None:
            self._shapes = shapes
            self._rows = [torch.empty((self.capacity, *s), dtype=torch.float32, device=self.device) for s in shapes]
        if shapes != self._shapes or id(sample) in self._slot_of or not self._free:
            return id(sample) in self._slot_of
        row = self._free.pop()
        for dst, src in zip(self._rows, fields):
            dst[row].copy_(src, non_blocking=True)
        self._slot_of[id(sample)] = row
        self._owner[row] = sample
        return True

    def evict(self, sample) -> None:
        row = self._slot_of.pop(id(sample), None)
        if row is not None:
            self._owner[row] = None
            self._free.append(row)

    def clear(self) -> None:
        self._slot_of.clear()
        self._owner = [None] * self.capacity
        self._free = list(range(self.capacity - 1, -1, -1))

    def gather(self, samples) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """(metrics [B,...], logs [B,...], labels [B,...]) on the device, rows in the order of `samples`; None when
        one of them is not in the pool."""
        try:
            rows = [self._slot_of[id(s)] for s in samples]
        except KeyError:
            return None
        if not rows:
            return None
        idx = torch.tensor(rows, dtype=torch.long)
        if self.device.type == "cuda":
            idx = idx.pin_memory().to(self.device, non_blocking=True)
        m, l, y = (r.index_select(0, idx) for r in self._rows)
        return m, l, y
