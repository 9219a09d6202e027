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
