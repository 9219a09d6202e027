"""Online learner: per-sample / per-batch update steps on MI355X.

Same classes, constructor arguments, return dictionaries and checkpoint
layout as the reference (ocean/models/training/online_learner.py:26-580).
The update arithmetic - forward, loss, backward, clip_grad_norm_(1.0), Adam -
runs in HIP kernels; the buffers and the learning-rate policy are host logic
reproduced as they are.  One device->host read per step (the three loss
scalars at once) replaces the reference's 3-6 .item() calls.
"""
from __future__ import annotations

import logging
import queue
import threading
from collections import defaultdict, deque
from datetime import datetime, timedelta
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.optim as optim

from ... import _hip
from ...configs import OCEANConfig
from ...data.data_types import DatasetSample
from ...optim import FusedAdam
from ...staging import BatchStager, ReplayPool
from ..components import DeviceGraph
from ..ocean_model import OCEANModel

logger = logging.getLogger(__name__)


class MemoryBuffer:
    """Bounded store of recent samples with an importance score each
    (online_learner.py:26-127): 'fifo', 'random' or 'importance' eviction;
    importance-weighted sampling without replacement."""

    def __init__(self, max_size: int = 1000, retention_strategy: str = "fifo"):
        self.max_size, self.retention_strategy = max_size, retention_strategy
        self.buffer: deque = deque(maxlen=max_size)
        self.importance_scores: deque = deque(maxlen=max_size)
        self.timestamps: deque = deque(maxlen=max_size)
        self.on_evict = None        # optional callable(sample): told which sample an eviction removed

    def add(self, sample: DatasetSample, importance_score: float = 1.0):
        if len(self.buffer) >= self.max_size:
            self._remove_sample()
        self.buffer.append(sample)
        self.importance_scores.append(importance_score)
        self.timestamps.append(datetime.now())

    def _drop_at(self, idx: int):
        if self.on_evict is not None:
            self.on_evict(self.buffer[idx])
        for name in ("buffer", "importance_scores", "timestamps"):
            items = list(getattr(self, name))
            items.pop(idx)
            setattr(self, name, deque(items, maxlen=self.max_size))

    def _remove_sample(self):
        if self.retention_strategy == "fifo":
            # the sample deque evicts its own oldest entry on append
            if self.on_evict is not None and len(self.buffer) > 0:
                self.on_evict(self.buffer[0])
            self.importance_scores.popleft()
            self.timestamps.popleft()
        elif self.retention_strategy == "random" and len(self.buffer) > 0:
            self._drop_at(int(np.random.randint(0, len(self.buffer))))
        elif self.retention_strategy == "importance" and len(self.importance_scores) > 0:
            self._drop_at(int(np.argmin(list(self.importance_scores))))

    def sample(self, batch_size: int) -> List[DatasetSample]:
        n = len(self.buffer)
        if n == 0:
            return []
        if batch_size >= n:
            return list(self.buffer)
        if self.retention_strategy == "importance":
            w = np.array(list(self.importance_scores))
            idx = np.random.choice(n, size=batch_size, replace=False, p=w / w.sum())
        else:
            idx = np.random.choice(n, size=batch_size, replace=False)
        return [self.buffer[i] for i in idx]

    def get_recent(self, time_window: timedelta) -> List[DatasetSample]:
        cutoff = datetime.now() - time_window
        return [s for s, t in zip(self.buffer, self.timestamps) if t >= cutoff]

    def __len__(self):
        return len(self.buffer)


class SlidingWindow:
    def __init__(self, window_size: int, step_size: int = 1):
        self.window_size, self.step_size = window_size, step_size
        self.window: deque = deque(maxlen=window_size)
        self.step_counter = 0

    def add(self, sample: DatasetSample) -> bool:
        self.window.append(sample)
        self.step_counter += 1
        return len(self.window) == self.window_size and self.step_counter % self.step_size == 0

    def get_window(self) -> List[DatasetSample]:
        return list(self.window)

    def reset(self):
        self.window.clear()
        self.step_counter = 0


class AdaptiveLearningRateScheduler:
    """Decay lr by `decay_factor` whenever, over a full window of losses, the
    newer half's mean is not below the older half's (online_learner.py:200-227)."""

    def __init__(self, initial_lr: float = 0.001, decay_factor: float = 0.95, adaptation_window: int = 100,
                 min_lr: float = 1e-6):
        self.initial_lr = self.current_lr = initial_lr
        self.decay_factor, self.adaptation_window, self.min_lr = decay_factor, adaptation_window, min_lr
        self.performance_history: deque = deque(maxlen=adaptation_window)
        self.lr_history: List[float] = []

    def step(self, performance_metric: float, optimizer: optim.Optimizer) -> float:
        self.performance_history.append(performance_metric)
        if len(self.performance_history) >= self.adaptation_window:
            hist, half = list(self.performance_history), self.adaptation_window // 2
            if np.mean(hist[-half:]) >= np.mean(hist[:half]):
                self.current_lr = max(self.current_lr * self.decay_factor, self.min_lr)
                for group in optimizer.param_groups:
                    group["lr"] = self.current_lr
        self.lr_history.append(self.current_lr)
        return self.current_lr

    def reset(self):
        self.current_lr = self.initial_lr
        self.performance_history.clear()


class _GraphedStep:
    """One update step (forward, loss, backward, clip + Adam) recorded as a HIP graph for a fixed batch shape and
    service graph, and replayed with one launch.  The recorded kernels keep their arguments, so everything that
    must change between steps lives in device memory: the inputs (copied into static buffers before a replay),
    the dropout epoch counter every mask-drawing kernel adds to its seed, and the optimizer's learning rate and
    step count.  The reference has no counterpart (it launches ~10^3 kernels per step from Python)."""

    def __init__(self, learner: "OnlineLearner", metrics, service_graph, logs, labels):
        model, opt = learner.model, learner.optimizer
        dev = model.device
        if opt._step_dev is None:
            opt.make_capturable(dev, learner._trained_params())
        self.metrics = metrics.detach().to(dev, torch.float32).clone()
        self.logs = logs.detach().to(dev, torch.float32).clone()
        self.labels = labels.detach().to(dev, torch.float32).clone()
        self.service_graph = service_graph
        self.epoch = _hip.seed_epoch(dev)
        self.generation = opt.generation           # recordings made before a load_state_dict point at replaced tensors
        clip = learner.config.training.gradient_clip_norm
        # the chunk table of the recorded optimizer step is staged through page-locked memory: allocate that buffer
        # now, outside the capture (the two eager steps before a recording left a gradient on every trained parameter)
        opt.reserve_table_host([p for g in opt.param_groups for p in g["params"] if p.grad is not None])
        opt.zero_grad(set_to_none=True)            # gradients are re-created inside the graph's memory pool (or live in the arena)
        opt.settle()
        one = learner._one()
        arena = learner._grad_arena()
        transport = getattr(learner.data_parallel, "transport", None)
        if transport is not None and hasattr(transport, "quiesce"):
            transport.quiesce()                        # no collective of the eager steps is still being watched
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _hip.capture(self.graph):
            self.epoch.add_(1)
            outputs = model(self.metrics, service_graph, self.logs)
            losses = model.compute_loss(outputs, self.labels)
            if arena is not None:
                arena.begin_step()                     # one fill node for every gradient of the step
            try:
                losses["total_loss"].backward(gradient=one)
            finally:
                if arena is not None:
                    arena.end_step()
            if learner.grad_sync is not None:
                learner.grad_sync(model)               # RCCL collectives: recorded as nodes of the graph
            opt.step(max_norm=clip if clip > 0 else 0.0)
            self.keys, self.loss_vec = learner._loss_vector(losses)
            if learner.data_parallel is not None:
                self.loss_vec = learner.data_parallel.reduce_losses(self.loss_vec)
        self.outputs = outputs
        # recording advanced the host-side step counters of the parameters it updates without running anything: undo
        # that, for those parameters only (a frozen component has state from earlier steps but is not in this step)
        self.params = [p for g in opt.param_groups for p in g["params"] if p.grad is not None]
        opt.bump_steps(self.params, -1)
        # buffers the recorded copy nodes and kernels read on every replay
        self._keep = (opt._table_host, opt._table, opt._sq)

    def replay(self, learner: "OnlineLearner", metrics, logs, labels) -> Dict[str, float]:
        self.metrics.copy_(metrics, non_blocking=True)
        self.logs.copy_(logs, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        learner.optimizer.sync_lr()               # a learning rate the host changed since the last step
        self.graph.replay()
        learner.optimizer.after_replay(self.params)
        learner.update_count += 1
        learner._before_readback()
        return dict(zip(self.keys, self.loss_vec.cpu().tolist()))


class OnlineLearner:
    def __init__(self, model: OCEANModel, config: OCEANConfig, optimizer: Optional[optim.Optimizer] = None):
        self.model, self.config = model, config
        self.optimizer = optimizer if optimizer is not None else FusedAdam(
            model.parameters(), lr=config.training.learning_rate, weight_decay=config.training.weight_decay)
        self.memory_buffer = MemoryBuffer(max_size=config.training.sliding_window_size, retention_strategy="importance")
        self.sliding_window = SlidingWindow(window_size=getattr(config.data, "sequence_length", 100),
                                            step_size=config.training.update_frequency)
        self.lr_scheduler = AdaptiveLearningRateScheduler(initial_lr=config.training.learning_rate)
        self.performance_history = defaultdict(list)
        self.update_count = 0
        self.last_update_time = datetime.now()
        self.data_queue: queue.Queue = queue.Queue(maxsize=1000)
        self.is_streaming = False
        self.stream_thread: Optional[threading.Thread] = None
        # data-parallel hooks (ocean.parallel.DataParallel sets both): grad_sync(model) runs after backward and before
        # the update; data_parallel knows whether its collectives can sit in a recorded step and how to weight a step in
        # which the ranks hold different numbers of rows
        self.grad_sync = None
        self.data_parallel = None
        # use_hip_graph: after two eager steps with the same batch shape and graph, process_tensors records the
        # step as a HIP graph and replays it from then on (single-GPU, FusedAdam, in-batch contrastive mode)
        self.use_hip_graph = False
        self.max_graphs = 4            # recorded steps kept (each owns its activations' memory); least recently used goes first
        self._graphed: Dict[Any, _GraphedStep] = {}
        # pinned_staging: process_batch collates host samples through page-locked buffers and reused device
        # tensors (ocean/staging.py) instead of three pageable torch.stack results; same values either way
        self.pinned_staging = True
        self._stager: Optional[BatchStager] = None
        self._prefetch_next: Optional[List[DatasetSample]] = None
        # device_replay: samples taken into the memory buffer are also kept in HBM (ocean/staging.py ReplayPool), so
        # experience_replay gathers its batch on the device instead of re-collating it from host tensors
        self.device_replay = True
        self._replay_pool: Optional[ReplayPool] = None
        self._mods = None
        self._prefetched = None
        self._graph_seen: Dict[Any, int] = defaultdict(int)
        self._replay_warned: set = set()
        self._seed_one: Optional[torch.Tensor] = None
        # grad_arena: the gradients of an update live in one persistent buffer (ocean._hip.GradArena) that is zeroed
        # with one fill before backward, instead of one fresh zero-filled allocation per component; their addresses
        # then never change, so the optimizer's chunk table is built once
        self.grad_arena = True
        self._arena: Optional[_hip.GradArena] = None

    # ------------------------------------------------------------------ the update step
    def _grad_arena(self) -> Optional[_hip.GradArena]:
        """The arena the coming backward pass writes its parameter gradients into (ocean._hip.step_arena)."""
        return _hip.step_arena(self)

    def _losses_to_host(self, losses: Dict[str, Any]) -> Dict[str, float]:
        """total / prediction / contrastive loss values with one device read."""
        keys, vec = self._loss_vector(losses)
        vals = []
        if keys:
            if self.data_parallel is not None:
                vec = self.data_parallel.reduce_losses(vec)      # every rank reports (and schedules on) the global means
            vals = vec.cpu().tolist()
        out = dict(zip(keys, vals))
        for k in ("total_loss", "prediction_loss", "contrastive_loss"):
            if k in losses and k not in out:
                out[k] = float(losses[k])
        return out

    @staticmethod
    def _loss_vector(losses: Dict[str, Any]):
        """(names, device vector) of the total / prediction / contrastive losses that are tensors: compute_loss's own
        [total, prediction, contrastive] vector when it left one, else the scalars stacked."""
        keys = [k for k in ("total_loss", "prediction_loss", "contrastive_loss") if torch.is_tensor(losses.get(k))]
        vec = getattr(losses, "vec", None)
        if vec is not None and keys[:2] == ["total_loss", "prediction_loss"]:
            return keys, vec.detach()[:len(keys)]
        if not keys:
            return keys, None
        return keys, torch.stack([losses[k].detach().float().reshape(()) for k in keys])

    def _one(self) -> torch.Tensor:
        """The backward pass's seed gradient (autograd would otherwise fill a fresh tensor with 1 on every step)."""
        dev = self.model.device
        if self._seed_one is None or self._seed_one.device != dev:
            self._seed_one = torch.ones((), device=dev, dtype=torch.float32)
        return self._seed_one

    def _train_mode(self) -> None:
        """model.train() as the reference calls it before every update (online_learner.py:304,375), skipped when
        every module already is in training mode: the recursive call walks ~200 modules and costs more host time
        than the kernels of a small update take to launch."""
        if not (self.model.training and all(m.training for m in self._modules_list())):
            self.model.train()

    def _modules_list(self):
        if self._mods is None:
            self._mods = list(self.model.modules())
        return self._mods

    def _weighted(self, total_loss: torch.Tensor) -> torch.Tensor:
        """Data parallel with unequal batches per rank: the local mean loss times world * local rows / global rows, so
        that the rank-averaged gradient is that of the mean over all rows (1 for equal batches)."""
        dp = self.data_parallel
        w = 1.0 if dp is None else dp.loss_weight()
        return total_loss if w == 1.0 else total_loss * w

    def _apply_update(self, total_loss: torch.Tensor) -> None:
        self.optimizer.zero_grad()
        loss = self._weighted(total_loss)
        arena = self._grad_arena()
        if arena is not None:
            arena.begin_step()
        try:
            if loss.dim() == 0 and loss.dtype == torch.float32 and loss.device == self.model.device:
                loss.backward(gradient=self._one())
            else:
                loss.backward()
        finally:
            if arena is not None:
                arena.end_step()
        if self.grad_sync is not None:
            self.grad_sync(self.model)
        clip = self.config.training.gradient_clip_norm
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.step(max_norm=clip if clip > 0 else 0.0)
        else:
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
            self.optimizer.step()
        self.update_count += 1

    def process_sample(self, sample: DatasetSample, update_model: bool = True) -> Dict[str, float]:
        self._train_mode()
        metrics, logs, label = sample.metrics.unsqueeze(0), sample.logs.unsqueeze(0), sample.label.unsqueeze(0)
        # the per-sample update is the latency-critical path of the streaming worker: with use_hip_graph it is one
        # graph launch, like the batch update
        host = self._graph_step(metrics, sample.graph, logs, label) if (update_model and self.use_hip_graph) else None
        if host is None:
            outputs = self.model(metrics, sample.graph, logs)
            losses = self.model.compute_loss(outputs, label.to(self.model.device, non_blocking=True))
            if update_model:
                self._apply_update(losses["total_loss"])
            host = self._losses_to_host(losses)
        importance_score = host["total_loss"]
        self._remember(sample, importance_score)
        if update_model:
            current_lr = self.lr_scheduler.step(host["total_loss"], self.optimizer)
            self.performance_history["loss"].append(host["total_loss"])
            self.performance_history["learning_rate"].append(current_lr)
        results = {"loss": host["total_loss"], "prediction_loss": host["prediction_loss"],
                   "importance_score": importance_score, "memory_buffer_size": len(self.memory_buffer),
                   "update_count": self.update_count}
        if "contrastive_loss" in host:
            results["contrastive_loss"] = host["contrastive_loss"]
        return results

    def process_batch(self, batch_samples: List[DatasetSample], prefetch: Optional[List[DatasetSample]] = None) -> Dict[str, float]:
        """One update from a list of samples (online_learner.py:362-431).  `prefetch` (not in the reference) names the
        list the caller will pass next: it is collated and copied to the device while the GPU works on this
        update, and a following process_batch(prefetch) call with that same list object starts on staged data."""
        if not batch_samples:
            return {}
        self._train_mode()
        service_graph = batch_samples[0].graph          # one graph for the batch (online_learner.py:382-383)
        staged = self._take_staged(batch_samples)
        if staged is not None:
            self._prefetch_next = prefetch
            try:
                self._stager.acquire(staged)
                return self.process_tensors(staged[0], service_graph, staged[1], staged[2])
            finally:
                self._prefetch_next = None
                self._stager.release(staged)
        metrics = torch.stack([s.metrics for s in batch_samples])
        logs = torch.stack([s.logs for s in batch_samples])
        labels = torch.stack([s.label for s in batch_samples])
        return self.process_tensors(metrics, service_graph, logs, labels)

    def process_standin(self, seq_len: int, service_graph) -> Dict[str, float]:
        """This rank's part in a data-parallel update for which it has no samples (sharded streams, uneven arrivals):
        a one-row batch of zeros runs through the same kernels and collectives so that every rank's calls stay
        matched; DataParallel.set_step_rows has told the wiring that the row counts as none (it adds nothing to the
        BatchNorm statistics, the memory bank or - its loss weight being 0 - the gradients)."""
        dev, m = self.model.device, self.model
        metrics = torch.zeros(1, seq_len, m.metrics_input_dim, device=dev)
        logs = torch.zeros(1, m.log_input_dim, device=dev)
        labels = torch.zeros(1, m.config.model.num_services, device=dev)
        return self.process_tensors(metrics, service_graph, logs, labels)

    def _take_staged(self, batch_samples):
        """Device tensors of this batch out of the pinned staging ring: the ones a previous call prefetched for
        this very list, or a fresh collate; None when staging does not apply."""
        held, self._prefetched = self._prefetched, None
        if held is not None and held[0] is batch_samples and len(batch_samples) == held[1]:
            return held[2]
        if not (self.pinned_staging and self.model.device.type == "cuda" and BatchStager.stageable(batch_samples)):
            return None
        if self._stager is None or self._stager.device != self.model.device:
            self._stager = BatchStager(self.model.device)
        return self._stager.collate(batch_samples)

    def _before_readback(self) -> None:
        """Called once per update after its last kernel is queued and before the host blocks on the loss values:
        the host is idle from here to the end of the GPU work, so the next batch is collated now."""
        nxt, self._prefetch_next = self._prefetch_next, None
        if nxt and self._stager is not None and BatchStager.stageable(nxt):
            staged = self._stager.collate(nxt)
            if staged is not None:
                self._prefetched = (nxt, len(nxt), staged)

    def process_tensors(self, metrics: torch.Tensor, service_graph, logs: torch.Tensor, labels: torch.Tensor) -> Dict[str, float]:
        """process_batch on already-collated tensors ([B,T,F], graph, [B,768], [B,S]); tensors that
        already live on the model's device are used in place (no host staging)."""
        self._train_mode()
        host = self._graph_step(metrics, service_graph, logs, labels) if self.use_hip_graph else None
        if host is None:
            outputs = self.model(metrics, service_graph, logs)
            losses = self.model.compute_loss(outputs, labels.to(self.model.device, non_blocking=True))
            self._apply_update(losses["total_loss"])
            self._before_readback()
            host = self._losses_to_host(losses)
        current_lr = self.lr_scheduler.step(host["total_loss"], self.optimizer)
        self.performance_history["batch_loss"].append(host["total_loss"])
        self.performance_history["learning_rate"].append(current_lr)
        return {"batch_loss": host["total_loss"], "batch_prediction_loss": host["prediction_loss"],
                "batch_size": int(metrics.shape[0]), "learning_rate": current_lr, "update_count": self.update_count}

    def _trained_params(self):
        """Parameters the next update will change: those that require a gradient and received one on the last step
        (never-used tensors such as the three unused cross-modal attention modules get none, ever)."""
        return [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad and p.grad is not None]

    def _replay_ineligible(self) -> Optional[str]:
        """Why this learner's update step cannot be recorded and replayed as a HIP graph, or None when it can."""
        if self.model.device.type != "cuda":
            return "the model is not on a GPU"
        if not isinstance(self.optimizer, FusedAdam):
            return "the optimizer is not ocean.optim.FusedAdam (its step is not capturable)"
        if len(self.optimizer.param_groups) != 1:
            return "the optimizer has several parameter groups"
        if self.grad_sync is not None and not getattr(self.data_parallel, "capturable", False):
            return "the gradient synchronisation hook's collectives cannot be captured (no RCCL transport)"
        if self.data_parallel is not None and getattr(self.data_parallel, "_rows_all", None) is not None:
            return "sharded streams: the ranks' row counts change from step to step"
        if self.model.contrastive_negatives != "in_batch" and not getattr(
                self.model.fusion_module.contrastive_module, "device_ring", False):
            return "memory-bank negatives with the ring pointer kept on the host"
        return None

    def _graph_step(self, metrics, service_graph, logs, labels) -> Optional[Dict[str, float]]:
        """Replay (or, on the third step with one signature, record) the HIP graph of this update; None when the
        step has to run eagerly (said once per reason in the log)."""
        why = self._replay_ineligible()
        if why is not None:
            if why not in self._replay_warned:
                self._replay_warned.add(why)
                logger.warning("use_hip_graph is set but the update step runs as separate launches: %s", why)
            return None
        adj = service_graph.adjacency_matrix
        # everything a recording bakes in: shapes, the graph's tensors, which components train, and the optimizer's
        # by-value hyper-parameters (the learning rate and the step count are read from device memory instead)
        grp = self.optimizer.param_groups[0]
        nf = service_graph.node_features
        # a tensor that REPLACES the adjacency or the node features starts again at version 0 under the same graph
        # object, so the storage addresses and shapes are part of the key as well (OCEANModel.device_graph keys the
        # device CSR the same way: both paths then pick up a new topology on the same step)
        key = (tuple(metrics.shape), tuple(logs.shape), tuple(labels.shape), id(service_graph)) + DeviceGraph.tensor_key(adj) + (
               nf.data_ptr(), nf._version, tuple(nf.shape), getattr(self.model, "trainable_version", 0),
               tuple(grp["betas"]), grp["eps"], grp["weight_decay"], self.config.training.gradient_clip_norm,
               self.model.contrastive_negatives, _hip.split_bf16())
        step = self._graphed.get(key)
        if step is not None and step.generation != self.optimizer.generation:
            self._drop_recordings()                # the optimizer state was reloaded: every recording is stale
            step = None
        if step is None:
            # (checked only when a step is about to be counted or recorded: the key holds the trainable set's version,
            # so step counts cannot drift apart under an existing recording without the key changing)
            if self.optimizer.common_step(self._trained_params()) is None:
                why = "the trained parameters have different Adam step counts (a component was frozen for a while)"
                if why not in self._replay_warned:
                    self._replay_warned.add(why)
                    logger.warning("use_hip_graph is set but the update step runs as separate launches: %s", why)
                return None
            self._graph_seen[key] = self._graph_seen.pop(key, 0) + 1      # re-inserted: most recently seen last
            while len(self._graph_seen) > 64:                  # bounded: the signature seen longest ago goes first
                self._graph_seen.pop(next(iter(self._graph_seen)))
            if self._graph_seen[key] <= 2:
                return None
            before = torch.cuda.current_stream(self.model.device)
            try:
                step = _GraphedStep(self, metrics, service_graph, logs, labels)
            except Exception as e:
                # The configurations known not to be capturable were turned away above, so this is a defect, and not
                # one to train through: a recording that dies half-way leaves PyTorch's capture bookkeeping (current
                # stream, random generator, allocator pool) in a state eager launches cannot rely on.  Put back what can
                # be put back, switch recording off for this learner, and say so.
                torch.cuda.set_stream(before)
                _hip.clear_last_error()
                self.use_hip_graph = False
                self.optimizer.zero_grad(set_to_none=True)
                raise RuntimeError("recording the update step as a HIP graph failed; use_hip_graph has been switched off for "
                                   "this learner, but the process should be restarted before training on") from e
            while len(self._graphed) >= max(1, self.max_graphs):
                old = self._graphed.pop(next(iter(self._graphed)))      # dicts keep insertion order; a hit re-inserts (below)
                self.optimizer.recycle_table_host(old._keep[0])
            self._graphed[key] = step
        else:
            self._graphed[key] = self._graphed.pop(key)           # mark as most recently used
        return step.replay(self, metrics, logs, labels)

    def _drop_recordings(self) -> None:
        for old in self._graphed.values():
            self.optimizer.recycle_table_host(old._keep[0])
        self._graphed.clear()

    def _remember(self, sample: DatasetSample, importance_score: float) -> None:
        """memory_buffer.add (online_learner.py:338), plus the sample's device copy when device_replay is on."""
        pool = self._replay_pool
        if self.device_replay and pool is None and self.model.device.type == "cuda":
            pool = self._replay_pool = ReplayPool(self.model.device, self.memory_buffer.max_size)
            self.memory_buffer.on_evict = pool.evict
        elif pool is not None and self.memory_buffer.on_evict != pool.evict:      # the buffer object was replaced
            pool.clear()
            self.memory_buffer.on_evict = pool.evict
        self.memory_buffer.add(sample, importance_score)         # evicts first when full, which frees a pool row
        if pool is not None and self.device_replay:
            pool.put(sample)

    def experience_replay(self, replay_batch_size: int = 32) -> Optional[Dict[str, float]]:
        """One update on a batch drawn from the memory buffer (online_learner.py:433-453).  The draw is the
        buffer's; with device_replay the batch is gathered from the samples' device copies."""
        if len(self.memory_buffer) < replay_batch_size:
            return None
        batch = self.memory_buffer.sample(replay_batch_size)
        staged = self._replay_pool.gather(batch) if (self.device_replay and self._replay_pool is not None) else None
        if staged is not None:
            results = self.process_tensors(staged[0], batch[0].graph, staged[1], staged[2])
        else:
            results = self.process_batch(batch)
        results["replay"] = True
        return results

    # ------------------------------------------------------------------ streaming thread
    def start_streaming(self):
        if self.is_streaming:
            logger.warning("Streaming already active")
            return
        self.is_streaming = True
        self.stream_thread = threading.Thread(target=self._streaming_worker, daemon=True)
        self.stream_thread.start()

    def stop_streaming(self):
        self.is_streaming = False
        if self.stream_thread:
            self.stream_thread.join(timeout=5.0)

    def add_streaming_sample(self, sample: DatasetSample):
        try:
            self.data_queue.put(sample, timeout=1.0)
        except queue.Full:
            logger.warning("Streaming queue full, dropping sample")

    def _streaming_worker(self):
        while self.is_streaming:
            try:
                sample = self.data_queue.get(timeout=1.0)
                results = self.process_sample(sample, update_model=True)
                if self.update_count % self.config.training.update_frequency == 0:
                    replay = self.experience_replay()
                    if replay:
                        logger.debug("Experience replay: %s", replay)
                if self.update_count % 100 == 0:
                    logger.info("Processed %d samples, current loss: %.4f", self.update_count, results["loss"])
            except queue.Empty:
                continue
            except Exception as e:          # keep the worker alive, as the reference does
                logger.error("Error in streaming worker: %s", e)

    # ------------------------------------------------------------------ bookkeeping
    def get_performance_summary(self) -> Dict[str, Any]:
        summary = {"total_updates": self.update_count, "memory_buffer_size": len(self.memory_buffer),
                   "is_streaming": self.is_streaming, "queue_size": self.data_queue.qsize(),
                   "current_lr": self.lr_scheduler.current_lr}
        for metric, values in self.performance_history.items():
            if values:
                summary[f"{metric}_mean"] = np.mean(values[-100:])
                summary[f"{metric}_std"] = np.std(values[-100:])
                summary[f"{metric}_trend"] = np.polyfit(range(len(values[-50:])), values[-50:], 1)[0] if len(values) >= 50 else 0
        return summary

    def save_checkpoint(self, filepath: str):
        # plain Python containers and floats only, so that load_checkpoint can use torch's restricted unpickler
        torch.save({"model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "update_count": int(self.update_count),
                    "performance_history": {k: [float(x) for x in v] for k, v in self.performance_history.items()},
                    "lr_scheduler_state": {"current_lr": float(self.lr_scheduler.current_lr),
                                           "performance_history": [float(x) for x in self.lr_scheduler.performance_history],
                                           "lr_history": [float(x) for x in self.lr_scheduler.lr_history]}}, filepath)

    def load_checkpoint(self, filepath: str):
        # weights_only: the file is unpickled with torch's restricted loader (tensors, dicts, lists, numbers, strings),
        # which is all save_checkpoint - and the reference's (online_learner.py:528-546) - ever writes
        ckpt = torch.load(filepath, map_location=self.model.device, weights_only=True)
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        self.update_count = ckpt["update_count"]
        for metric, values in ckpt["performance_history"].items():
            self.performance_history[metric] = values
        st = ckpt.get("lr_scheduler_state")
        if st:
            self.lr_scheduler.current_lr = st["current_lr"]
            self.lr_scheduler.performance_history = deque(st["performance_history"], maxlen=self.lr_scheduler.adaptation_window)
            self.lr_scheduler.lr_history = st["lr_history"]

    def reset(self):
        self.memory_buffer = MemoryBuffer(max_size=self.config.training.sliding_window_size, retention_strategy="importance")
        self._replay_pool = None
        self.sliding_window.reset()
        self.lr_scheduler.reset()
        self.performance_history.clear()
        self.update_count = 0
