"""Batch trainer: LossFunction and the epoch step on MI355X.

LossFunction reproduces the arithmetic of the reference's
(ocean/models/training/trainer.py:29-120): BCE + 0.1 * sum of InfoNCE terms
+ 0.05 * MSE(temporal_t, temporal_{t-1}) + 0.02 * mean |off-diagonal cosine of
the fused batch|, every term a HIP kernel.  Trainer keeps the reference's
constructor and train() / _train_epoch / _validate_epoch / checkpoint methods
(trainer.py:245-645) for any loader that yields the reference's batch
dictionaries ('metrics', 'logs', 'labels', 'graphs').

The reference's own loop cannot finish an epoch (it calls .item() on Python
floats, trainer.py:91,430, and reads a 'val_loss' key it never writes,
:358); here scalar terms are handled and validation is monitored on
'val_total_loss'.  Plotting and wandb logging are host tooling outside the
model path and are not provided.
"""
from __future__ import annotations

import json
import logging
from collections import defaultdict
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from ... import _hip, ops
from ...configs import OCEANConfig
from ...optim import FusedAdam
from ..components import DeviceGraph
from ..ocean_model import OCEANModel

logger = logging.getLogger(__name__)


class LossFunction(nn.Module):
    def __init__(self, prediction_weight: float = 1.0, contrastive_weight: float = 0.1, consistency_weight: float = 0.05,
                 diversity_weight: float = 0.02):
        super().__init__()
        self.prediction_weight, self.contrastive_weight = prediction_weight, contrastive_weight
        self.consistency_weight, self.diversity_weight = consistency_weight, diversity_weight
        # data-parallel hook: [B_loc, E] -> [B_all, E] keeping autograd (backward: reduce-scatter).  The diversity term
        # couples every pair of rows of the batch (trainer.py:97-110), so N ranks evaluate it on the gathered rows; the
        # other terms are means over rows and need no exchange
        self.gather_rows = None

    def forward(self, predictions: Dict[str, Any], targets: torch.Tensor, temporal_features: Optional[torch.Tensor] = None,
                prev_temporal_features: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        losses: Dict[str, Any] = {}
        probs = predictions["root_cause_probs"]
        pred_loss = ops.binary_cross_entropy(probs, targets.to(probs.device))
        losses["prediction_loss"] = pred_loss
        total_c: Any = 0.0
        for name, value in predictions.get("contrastive_losses", {}).items():
            losses[f"contrastive_{name}"] = value
            total_c = total_c + value
        losses["total_contrastive_loss"] = total_c
        consistency: Any = 0.0
        if temporal_features is not None and prev_temporal_features is not None:
            consistency = ops.mse_loss(temporal_features, prev_temporal_features)
        losses["consistency_loss"] = consistency
        diversity: Any = 0.0
        inter = predictions.get("intermediate")
        if inter is not None and "fused_representation" in inter:
            fused = inter["fused_representation"]
            diversity = ops.diversity_loss(fused if self.gather_rows is None else self.gather_rows(fused))
        losses["diversity_loss"] = diversity
        losses["total_loss"] = (self.prediction_weight * pred_loss + self.contrastive_weight * total_c +
                                self.consistency_weight * consistency + self.diversity_weight * diversity)
        return losses


class EarlyStopping:
    def __init__(self, patience: int = 10, min_delta: float = 1e-4, monitor: str = "val_loss", mode: str = "min",
                 restore_best_weights: bool = True):
        self.patience, self.monitor, self.mode, self.restore_best_weights = patience, monitor, mode, restore_best_weights
        self.min_delta = -min_delta if mode == "min" else min_delta
        self.monitor_op = np.less if mode == "min" else np.greater
        self.wait = self.stopped_epoch = 0
        self.best_value: Optional[float] = None
        self.best_weights = None

    def _snapshot(self, model: nn.Module):
        return {k: v.detach().clone() for k, v in model.state_dict().items()}

    def __call__(self, current_value: float, model: nn.Module) -> bool:
        if self.best_value is None or self.monitor_op(current_value - self.min_delta, self.best_value):
            first = self.best_value is None
            self.best_value, self.best_weights = current_value, self._snapshot(model)
            if not first:
                self.wait = 0
            return False
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = self.wait
            if self.restore_best_weights and self.best_weights:
                model.load_state_dict(self.best_weights)
            return True
        return False


class LearningRateScheduler:
    def __init__(self, optimizer: optim.Optimizer, scheduler_type: str = "cosine", **kw):
        self.optimizer, self.scheduler_type = optimizer, scheduler_type
        sched = optim.lr_scheduler
        if scheduler_type == "cosine":
            self.scheduler = sched.CosineAnnealingLR(optimizer, T_max=kw.get("T_max", 100), eta_min=kw.get("eta_min", 1e-6))
        elif scheduler_type == "step":
            self.scheduler = sched.StepLR(optimizer, step_size=kw.get("step_size", 30), gamma=kw.get("gamma", 0.1))
        elif scheduler_type == "plateau":
            self.scheduler = sched.ReduceLROnPlateau(optimizer, mode=kw.get("mode", "min"), factor=kw.get("factor", 0.5),
                                                     patience=kw.get("patience", 10), min_lr=kw.get("min_lr", 1e-6))
        elif scheduler_type == "exponential":
            self.scheduler = sched.ExponentialLR(optimizer, gamma=kw.get("gamma", 0.95))
        else:
            raise ValueError(f"Unknown scheduler type: {scheduler_type}")

    def step(self, metric: Optional[float] = None):
        if self.scheduler_type == "plateau":
            if metric is not None:
                self.scheduler.step(metric)
        else:
            self.scheduler.step()

    def get_last_lr(self) -> List[float]:
        return self.scheduler.get_last_lr()


class _GraphedTrainStep:
    """One Trainer.train_step (forward with intermediates, LossFunction, backward, gradient collectives, clip + Adam)
    recorded as a HIP graph for fixed batch shapes and a fixed service graph, and replayed with one launch.  What changes
    between steps lives in device memory: the inputs (copied in before a replay), the previous batch's temporal features
    (the consistency term's second operand: a tensor of the recording that each replay reads and then overwrites with this
    batch's), the dropout epoch, the optimizer's learning rate and step count.  The reference has no counterpart."""

    def __init__(self, trainer: "Trainer", batch: Dict[str, Any]):
        model, opt = trainer.model, trainer.optimizer
        dev = trainer.device
        params = [p for g in opt.param_groups for p in g["params"] if p.requires_grad and p.grad is not None]
        if opt._step_dev is None:
            opt.make_capturable(dev, params)
        f32 = lambda t: t.detach().to(dev, torch.float32).clone()
        self.metrics, self.logs, self.labels = f32(batch["metrics"]), f32(batch["logs"]), f32(batch["labels"])
        self.service_graph = batch["graphs"][0]
        self.prev = trainer.prev_temporal_features.detach().clone()
        self.epoch = _hip.seed_epoch(dev)
        self.generation = opt.generation
        clip = trainer.config.training.gradient_clip_norm
        opt.reserve_table_host([p for g in opt.param_groups for p in g["params"] if p.grad is not None])
        opt.zero_grad(set_to_none=True)
        opt.settle()
        arena = _hip.step_arena(trainer)
        transport = getattr(trainer.data_parallel, "transport", None)
        if transport is not None and hasattr(transport, "quiesce"):
            transport.quiesce()                        # no collective of the eager steps is still being watched
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _hip.capture(self.graph):
            self.epoch.add_(1)
            out = model(self.metrics, self.service_graph, self.logs, return_intermediate=True)
            temporal = out["intermediate"]["temporal_features"]
            losses = trainer.loss_function(out, self.labels, temporal, self.prev)
            if arena is not None:
                arena.begin_step()
            try:
                losses["total_loss"].backward()
            finally:
                if arena is not None:
                    arena.end_step()
            if trainer.grad_sync is not None:
                trainer.grad_sync(model)
            opt.step(max_norm=clip if clip > 0 else 0.0)
            self.prev.copy_(temporal.detach())
            self.keys = [k for k, v in losses.items() if torch.is_tensor(v)]
            self.loss_vec = torch.stack([losses[k].detach().float().reshape(()) for k in self.keys])
            self.consts = {k: float(v) for k, v in losses.items() if not torch.is_tensor(v)}
        self.params = [p for g in opt.param_groups for p in g["params"] if p.grad is not None]
        opt.bump_steps(self.params, -1)           # recording advanced the host-side step counters without running anything
        self._keep = (opt._table_host, opt._table, opt._sq)

    def replay(self, trainer: "Trainer", batch: Dict[str, Any]) -> Dict[str, float]:
        self.metrics.copy_(batch["metrics"], non_blocking=True)
        self.logs.copy_(batch["logs"], non_blocking=True)
        self.labels.copy_(batch["labels"], non_blocking=True)
        if trainer.prev_temporal_features is not self.prev:
            self.prev.copy_(trainer.prev_temporal_features, non_blocking=True)
        trainer.optimizer.sync_lr()
        self.graph.replay()
        trainer.optimizer.after_replay(self.params)
        trainer.prev_temporal_features = self.prev
        host = dict(zip(self.keys, self.loss_vec.cpu().tolist()))
        host.update(self.consts)
        return host


class Trainer:
    def __init__(self, model: OCEANModel, config: OCEANConfig, data_loader: Any, optimizer: Optional[optim.Optimizer] = None,
                 device: Optional[torch.device] = None, use_wandb: bool = False):
        if use_wandb:
            raise NotImplementedError("wandb logging is not part of the MI355X model path")
        self.model, self.config, self.data_loader = model, config, data_loader
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.model.to(self.device)
        self.optimizer = optimizer if optimizer is not None else FusedAdam(
            model.parameters(), lr=config.training.learning_rate, weight_decay=config.training.weight_decay)
        self.loss_function = LossFunction(prediction_weight=config.training.prediction_loss_weight,
                                          contrastive_weight=config.training.contrastive_loss_weight)
        self.early_stopping = EarlyStopping(patience=config.training.early_stopping_patience,
                                            min_delta=config.training.early_stopping_delta)
        self.lr_scheduler = LearningRateScheduler(self.optimizer, scheduler_type=config.training.scheduler)
        self.current_epoch = 0
        self.train_history, self.val_history = defaultdict(list), defaultdict(list)
        self.best_val_loss = float("inf")
        self.training_start_time: Optional[datetime] = None
        self.online_learner = None
        self.prev_temporal_features: Optional[torch.Tensor] = None
        self.grad_sync = None                      # data-parallel hooks, see ocean.parallel.DataParallel
        self.data_parallel = None
        # use_hip_graph: after two eager steps with the same batch shapes and service graph, train_step records the step
        # as a HIP graph and replays it (fused optimizer, one parameter group; a batch of another shape - the ragged
        # last one of an epoch - and the step after it run as ordinary launches).  grad_arena: as in OnlineLearner.
        self.use_hip_graph = False
        self.max_graphs = 2
        self.grad_arena = True
        self._arena = None
        self._graphed: Dict[Any, _GraphedTrainStep] = {}
        self._graph_seen: Dict[Any, int] = {}

    # ------------------------------------------------------------------ one batch
    def _move_batch_to_device(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def _batch_losses(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        out = self.model(batch["metrics"], batch["graphs"][0], batch["logs"], return_intermediate=True)
        temporal = out["intermediate"]["temporal_features"]
        prev = self.prev_temporal_features
        if prev is not None and prev.shape != temporal.shape:     # a ragged last batch has no predecessor of its shape
            prev = None
        losses = self.loss_function(out, batch["labels"], temporal, prev)
        return out, losses

    @staticmethod
    def _to_host(losses: Dict[str, Any]) -> Dict[str, float]:
        keys = [k for k, v in losses.items() if torch.is_tensor(v)]
        vals = torch.stack([losses[k].detach().float().reshape(()) for k in keys]).cpu().tolist() if keys else []
        host = dict(zip(keys, vals))
        host.update({k: float(v) for k, v in losses.items() if not torch.is_tensor(v)})
        return host

    def _graph_step(self, batch: Dict[str, Any]) -> Optional[Dict[str, float]]:
        """Replay (or, on the third step with one signature, record) the HIP graph of this step; None when it has to run
        as ordinary launches."""
        opt = self.optimizer
        if (self.device.type != "cuda" or not isinstance(opt, FusedAdam) or len(opt.param_groups) != 1
                or (self.grad_sync is not None and not getattr(self.data_parallel, "capturable", False))
                or not getattr(self.model.fusion_module.contrastive_module, "device_ring", True)
                and self.model.contrastive_negatives != "in_batch"):
            return None
        metrics, logs, labels, sg = batch["metrics"], batch["logs"], batch["labels"], batch["graphs"][0]
        prev = self.prev_temporal_features
        if prev is None or prev.shape[0] != metrics.shape[0]:
            return None                    # no consistency term on this step (first batch, or after a ragged one)
        grp, adj, nf = opt.param_groups[0], sg.adjacency_matrix, sg.node_features
        lf = self.loss_function
        key = (tuple(metrics.shape), tuple(logs.shape), tuple(labels.shape), id(sg)) + DeviceGraph.tensor_key(adj) + (
               nf.data_ptr(), nf._version, tuple(nf.shape), getattr(self.model, "trainable_version", 0), tuple(grp["betas"]), grp["eps"],
               grp["weight_decay"], self.config.training.gradient_clip_norm, self.model.contrastive_negatives,
               lf.prediction_weight, lf.contrastive_weight, lf.consistency_weight, lf.diversity_weight, _hip.split_bf16())
        step = self._graphed.get(key)
        if step is not None and step.generation != opt.generation:
            self._graphed.clear()                  # the optimizer state was reloaded: every recording is stale
            step = None
        if step is None:
            trained = [p for g in opt.param_groups for p in g["params"] if p.requires_grad and p.grad is not None]
            if opt.common_step(trained) is None:
                return None
            n = self._graph_seen.pop(key, 0) + 1
            self._graph_seen[key] = n
            while len(self._graph_seen) > 64:
                self._graph_seen.pop(next(iter(self._graph_seen)))
            if n <= 2:
                return None
            before = torch.cuda.current_stream(self.device)
            try:
                step = _GraphedTrainStep(self, batch)
            except Exception as e:
                torch.cuda.set_stream(before)
                _hip.clear_last_error()
                self.use_hip_graph = False
                opt.zero_grad(set_to_none=True)
                raise RuntimeError("recording the training step as a HIP graph failed; use_hip_graph has been switched off for "
                                   "this trainer, but the process should be restarted before training on") from e
            while len(self._graphed) >= max(1, self.max_graphs):
                old = self._graphed.pop(next(iter(self._graphed)))
                opt.recycle_table_host(old._keep[0])
        else:
            self._graphed.pop(key)
        self._graphed[key] = step
        return step.replay(self, batch)

    def train_step(self, batch: Dict[str, Any]) -> Dict[str, float]:
        """Forward, LossFunction, backward, clip, optimizer step on one batch dictionary (trainer.py:395-434)."""
        self.model.train()
        batch = self._move_batch_to_device(batch)
        if self.use_hip_graph:
            host = self._graph_step(batch)
            if host is not None:
                return host
        out, losses = self._batch_losses(batch)
        self.optimizer.zero_grad()
        arena = _hip.step_arena(self)
        if arena is not None:
            arena.begin_step()
        try:
            losses["total_loss"].backward()
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
        self.prev_temporal_features = out["intermediate"]["temporal_features"].detach()
        return self._to_host(losses)

    def _train_epoch(self, train_loader) -> Dict[str, float]:
        sums = defaultdict(list)
        for batch in train_loader:
            for k, v in self.train_step(batch).items():
                sums[k].append(v)
        return {f"train_{k}": float(np.mean(v)) for k, v in sums.items()}

    def _validate_epoch(self, val_loader) -> Dict[str, float]:
        self.model.eval()
        sums = defaultdict(list)
        with torch.no_grad():
            for batch in val_loader:
                _, losses = self._batch_losses(self._move_batch_to_device(batch))
                for k, v in self._to_host(losses).items():
                    sums[k].append(v)
        return {f"val_{k}": float(np.mean(v)) for k, v in sums.items()}

    # ------------------------------------------------------------------ epochs
    def train(self, num_epochs: Optional[int] = None) -> Dict[str, Any]:
        num_epochs = self.config.training.num_epochs if num_epochs is None else num_epochs
        self.training_start_time = datetime.now()
        train_loader, val_loader, _ = self.data_loader.get_all_loaders()
        try:
            for epoch in range(num_epochs):
                self.current_epoch = epoch
                tm = self._train_epoch(train_loader)
                vm = self._validate_epoch(val_loader)
                val_loss = vm.get("val_total_loss", float("nan"))
                self.lr_scheduler.step(val_loss)
                for k, v in tm.items():
                    self.train_history[k].append(v)
                for k, v in vm.items():
                    self.val_history[k].append(v)
                self.best_val_loss = min(self.best_val_loss, val_loss)
                if self.early_stopping(val_loss, self.model):
                    logger.info("Early stopping at epoch %d", epoch)
                    break
                if (epoch + 1) % self.config.training.save_checkpoint_every == 0:
                    self._save_checkpoint(epoch, val_loss)
        except KeyboardInterrupt:
            logger.info("Training interrupted by user")
        finally:
            self._save_final_model()
        return {"train_history": dict(self.train_history), "val_history": dict(self.val_history)}

    # ------------------------------------------------------------------ checkpoints
    def _checkpoint_dict(self, epoch: int, val_loss: float) -> Dict[str, Any]:
        return {"epoch": epoch, "model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                "scheduler_state_dict": self._plain(self.lr_scheduler.scheduler.state_dict()), "val_loss": float(val_loss),
                "train_history": self._plain(dict(self.train_history)), "val_history": self._plain(dict(self.val_history)),
                "config": self._plain(self.config.to_dict())}

    @staticmethod
    def _plain(obj):
        """Tensors stay; numpy scalars become Python numbers, tuples / other containers become lists / dicts, and anything
        else its repr - so that the checkpoint loads under torch.load(weights_only=True)."""
        if torch.is_tensor(obj) or obj is None or isinstance(obj, (bool, int, float, str)):
            return obj
        if isinstance(obj, np.generic):
            return obj.item()
        if isinstance(obj, dict):
            return {(k if isinstance(k, (str, int, float, bool)) else str(k)): Trainer._plain(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple, set)):
            return [Trainer._plain(v) for v in obj]
        return repr(obj)

    def _save_checkpoint(self, epoch: int, val_loss: float):
        d = Path(self.config.training.checkpoint_dir)
        d.mkdir(parents=True, exist_ok=True)
        torch.save(self._checkpoint_dict(epoch, val_loss), d / f"checkpoint_epoch_{epoch + 1}.pt")

    def _save_final_model(self):
        d = Path(self.config.training.checkpoint_dir)
        d.mkdir(parents=True, exist_ok=True)
        torch.save({"model_state_dict": self.model.state_dict(), "config": self.config.to_dict(),
                    "training_completed": datetime.now().isoformat(), "total_epochs": self.current_epoch + 1,
                    "best_val_loss": self.best_val_loss}, d / "final_model.pt")
        with open(d / "training_history.json", "w") as f:
            json.dump({"train_history": dict(self.train_history), "val_history": dict(self.val_history)}, f, indent=2)

    def load_checkpoint(self, checkpoint_path: str):
        # torch's restricted unpickler: checkpoints hold tensors, numbers, strings and plain containers only
        ckpt = torch.load(checkpoint_path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if "scheduler_state_dict" in ckpt:
            self.lr_scheduler.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.current_epoch = ckpt.get("epoch", 0)
        for k, v in ckpt.get("train_history", {}).items():
            self.train_history[k] = v
        for k, v in ckpt.get("val_history", {}).items():
            self.val_history[k] = v
