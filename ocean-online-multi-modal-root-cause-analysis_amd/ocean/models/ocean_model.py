"""OCEANModel: the per-step model hot path on MI355X behind the reference API.

Same constructor, forward signature, output dictionary, compute_loss,
predict_root_cause and state_dict keys as the reference's OCEANModel
(ocean/models/ocean_model.py:22-452).  Every tensor operation of the forward
and backward pass runs in libocean_hip.so; this file only validates shapes,
moves host inputs to the model's device and wires the five component drivers
(DilatedCNN, GraphNeuralNetwork, MultiFactorAttention, GraphFusionModule,
prediction head) together through torch.autograd.
"""
from __future__ import annotations

import ctypes
import logging
import os
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from .. import _hip, ops
from ..configs import OCEANConfig
from ..data.data_types import ServiceGraph
from .components import DeviceGraph, DilatedCNN, GraphFusionModule, GraphNeuralNetwork, MultiFactorAttention

logger = logging.getLogger(__name__)


class _HeadFunction(torch.autograd.Function):
    """prediction_head (3 Linear, ReLU, Dropout 0.1) + sigmoid -> (logits, probs)."""

    @staticmethod
    def forward(ctx, x, training, p_drop, *params):
        L = _hip.lib()
        sh = _hip.HeadShape()
        sh.B, sh.E = x.shape
        sh.h1, sh.h2, sh.S = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        sh.training, sh.dropout = int(training), (p_drop if training else 0.0)
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(x.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        ws = torch.empty(L.ocean_head_workspace(ctypes.byref(sh)), device=x.device, dtype=torch.float32)
        logits = torch.empty(sh.B, sh.S, device=x.device, dtype=torch.float32)
        probs = torch.empty_like(logits)
        _hip.check(L.ocean_head_fwd(ctypes.byref(sh), _hip.dev_ptr(x, "fused"), _hip.ptr_table(params, "head param"), _hip.dev_ptr(ws),
                                    ws.numel(), _hip.dev_ptr(logits), _hip.dev_ptr(probs), _hip.stream()), "ocean_head_fwd")
        ctx.sh = sh
        ctx.save_for_backward(x, ws, probs, *params)
        return logits, probs

    @staticmethod
    def backward(ctx, dlogits, dprobs):
        L = _hip.lib()
        x, ws, probs, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_head_bwd_workspace(ctypes.byref(sh)), device=x.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        c = lambda g: None if g is None else g.contiguous()
        _hip.check(L.ocean_head_bwd(ctypes.byref(sh), _hip.dev_ptr(x), _hip.ptr_table(params, "head param"), _hip.dev_ptr(ws),
                                    _hip.dev_ptr(wb), wb.numel(), _hip.dev_ptr(probs), _hip.dev_ptr(c(dlogits)), _hip.dev_ptr(c(dprobs)),
                                    _hip.ptr_table(grads, "head grad"), _hip.dev_ptr(dx), _hip.stream()), "ocean_head_bwd")
        return (dx, None, None) + tuple(g if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads))


class _GraphedForward:
    """One eval-mode forward pass (no gradients, no dropout, BatchNorm on its running statistics) recorded as a HIP graph
    for fixed input shapes and a fixed service graph, and replayed with one launch.  The recorded kernels read the
    parameters and buffers where they live, so updates made by training between two predictions are seen; the inputs are
    copied into the recording's own tensors before a replay and the outputs are copied out after it.  The reference has
    no counterpart (ocean_model.py:272-332 launches every operator from Python)."""

    def __init__(self, model: "OCEANModel", metrics, service_graph, logs):
        dev = model.device
        self.metrics = metrics.detach().to(dev, torch.float32).clone()
        self.logs = None if logs is None else logs.detach().to(dev, torch.float32).clone()
        self.service_graph = service_graph
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _hip.capture(self.graph), torch.no_grad():
            self.outputs = model._forward_impl(self.metrics, service_graph, self.logs, False)

    def replay(self, metrics, logs) -> Dict[str, Any]:
        self.metrics.copy_(metrics, non_blocking=True)
        if self.logs is not None:
            self.logs.copy_(logs, non_blocking=True)
        self.graph.replay()
        probs = self.outputs["root_cause_probs"].clone()
        return {"root_cause_logits": self.outputs["root_cause_logits"].clone(), "root_cause_probs": probs, "prediction": probs}


class OCEANModel(nn.Module):
    def __init__(self, config: OCEANConfig):
        super().__init__()
        self.config = config
        self.metrics_input_dim = getattr(config.data, "metrics_features", 12)
        self.log_input_dim = getattr(config.data, "log_embedding_dim", 768)
        self.graph_input_dim = getattr(config.data, "graph_features", 12)
        m = config.model
        self.temporal_dim = self.spatial_dim = m.hidden_dim
        self.attention_dim, self.fusion_dim = m.attention_dim, m.fusion_projection_dim

        self.temporal_encoder = DilatedCNN(input_dim=self.metrics_input_dim, channels=m.dcnn_channels,
                                           kernel_size=m.dcnn_kernel_size, dilation_rates=m.dcnn_dilation_rates,
                                           dropout=m.dcnn_dropout, output_dim=self.temporal_dim)
        self.spatial_encoder = GraphNeuralNetwork(input_dim=self.graph_input_dim, hidden_dim=self.spatial_dim,
                                                  num_layers=m.gnn_num_layers, num_heads=m.gnn_num_heads,
                                                  dropout=m.gnn_dropout, conv_type="gat", pooling="mean", batch_norm=False)
        self.attention_module = MultiFactorAttention(temporal_dim=self.temporal_dim, spatial_dim=self.spatial_dim,
                                                     log_dim=self.log_input_dim, attention_dim=self.attention_dim,
                                                     num_heads=m.attention_num_heads, dropout=m.attention_dropout,
                                                     fusion_strategy="concatenate")
        self.fusion_module = GraphFusionModule(temporal_dim=self.attention_dim, spatial_dim=self.attention_dim,
                                               log_dim=self.attention_dim, fusion_dim=self.fusion_dim,
                                               projection_dim=self.fusion_dim // 2, temperature=m.fusion_temperature,
                                               fusion_strategy="attention")
        self.prediction_head = nn.Sequential(
            nn.Linear(self.fusion_dim, self.fusion_dim // 2), nn.ReLU(), nn.Dropout(0.1),
            nn.Linear(self.fusion_dim // 2, self.fusion_dim // 4), nn.ReLU(), nn.Dropout(0.1),
            nn.Linear(self.fusion_dim // 4, m.num_services))
        self.output_activation = nn.Sigmoid()
        self.prediction_loss_weight = config.training.prediction_loss_weight
        self.contrastive_loss_weight = config.training.contrastive_loss_weight
        # overlap_encoders: run the graph encoder on a second HIP stream next to the dilated CNN (same results).
        # Off by default (OCEAN_OVERLAP_ENCODERS=1 turns it on for measurements): on the 256-service / batch-1024 step
        # it costs 0.3 ms (10.77 ms against 10.50; 10.72 with the second stream at the lowest priority) - the gather
        # kernels take compute units and L2 from the matrix kernels for longer than they save.
        self.overlap_encoders = os.environ.get("OCEAN_OVERLAP_ENCODERS", "0") == "1"
        self._side_streams: Dict[torch.device, Any] = {}
        # use_hip_graph: an eval-mode forward under torch.no_grad() (predict_root_cause, get-style calls) is recorded as a
        # HIP graph after two eager calls with the same input shapes and service graph, and replayed from then on
        self.use_hip_graph = False
        self.max_graphs = 4
        self._graphed_fwd: Dict[Any, _GraphedForward] = {}
        self._graph_fwd_seen: Dict[Any, int] = {}

    def __getstate__(self):
        """Copies and pickles of the module carry parameters, buffers and settings, not this process's device-side
        caches (recorded forward passes, side streams, the module list)."""
        state = dict(self.__dict__)
        state["_graphed_fwd"], state["_graph_fwd_seen"], state["_side_streams"] = {}, {}, {}
        state.pop("_all_modules", None)
        return state

    # ------------------------------------------------------------------ helpers
    @property
    def device(self) -> torch.device:
        return self.prediction_head[0].weight.device

    @property
    def contrastive_negatives(self) -> str:
        return self.fusion_module.contrastive_module.negatives

    @contrastive_negatives.setter
    def contrastive_negatives(self, mode: str) -> None:
        if mode not in ("bank", "in_batch"):
            raise ValueError("contrastive_negatives must be 'bank' or 'in_batch'")
        self.fusion_module.contrastive_module.negatives = mode

    def device_graph(self, service_graph: ServiceGraph) -> DeviceGraph:
        """Device CSR of the service graph, built once per (graph object,
        adjacency version, device) and cached on the graph."""
        adj = service_graph.adjacency_matrix
        key = DeviceGraph.tensor_key(adj) + (str(self.device),)
        cache = getattr(service_graph, "_ocean_device_graph", None)
        if cache is None or cache[0] != key:
            g = DeviceGraph.from_adjacency(adj, self.device)
            nf = service_graph.node_features.detach().to(self.device, torch.float32).contiguous()
            cache = (key, g, nf, (service_graph.node_features.data_ptr(), service_graph.node_features._version))
            service_graph._ocean_device_graph = cache
        nfk = (service_graph.node_features.data_ptr(), service_graph.node_features._version)
        if cache[3] != nfk:
            nf = service_graph.node_features.detach().to(self.device, torch.float32).contiguous()
            cache = (cache[0], cache[1], nf, nfk)
            service_graph._ocean_device_graph = cache
        return cache[1], cache[2]

    def _encoder_stream(self, dev: torch.device) -> "torch.cuda.Stream":
        st = self._side_streams.get(dev)
        if st is None:
            st = torch.cuda.Stream(dev)
            self._side_streams[dev] = st
        return st

    def _head_params(self):
        h = self.prediction_head
        return [h[0].weight, h[0].bias, h[3].weight, h[3].bias, h[6].weight, h[6].bias]

    # ------------------------------------------------------------------ forward
    def forward(self, metrics: torch.Tensor, service_graph: ServiceGraph, logs: Optional[torch.Tensor],
                return_intermediate: bool = False) -> Dict[str, Any]:
        if (self.use_hip_graph and not self.training and not torch.is_grad_enabled() and not return_intermediate
                and not self.overlap_encoders and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing()):
            out = self._replayed_forward(metrics, service_graph, logs)
            if out is not None:
                return out
        return self._forward_impl(metrics, service_graph, logs, return_intermediate)

    def _replayed_forward(self, metrics, service_graph, logs) -> Optional[Dict[str, Any]]:
        """Replay (or, on the third call with one signature, record) the HIP graph of this eval-mode forward; None
        while the signature is still new (the call then runs as ordinary launches)."""
        mods = self.__dict__.get("_all_modules")
        if mods is None:
            mods = self.__dict__["_all_modules"] = list(self.modules())
        if any(m.training for m in mods):
            return None                       # a submodule left in training mode: dropout / batch statistics would be live
        adj, nf = service_graph.adjacency_matrix, service_graph.node_features
        sentinel = self.prediction_head[0].weight
        key = (tuple(metrics.shape), None if logs is None else tuple(logs.shape), id(service_graph)) + DeviceGraph.tensor_key(adj) + (
               nf.data_ptr(), nf._version, tuple(nf.shape), sentinel.data_ptr(), str(sentinel.device), _hip.split_bf16())
        rec = self._graphed_fwd.get(key)
        if rec is None:
            n = self._graph_fwd_seen.pop(key, 0) + 1
            self._graph_fwd_seen[key] = n
            while len(self._graph_fwd_seen) > 64:
                self._graph_fwd_seen.pop(next(iter(self._graph_fwd_seen)))
            if n <= 2:
                return None
            before = torch.cuda.current_stream(self.device)
            try:
                rec = _GraphedForward(self, metrics, service_graph, logs)
            except Exception as e:
                torch.cuda.set_stream(before)
                _hip.clear_last_error()
                self.use_hip_graph = False
                raise RuntimeError("recording the forward pass as a HIP graph failed; use_hip_graph has been switched off for "
                                   "this model, but the process should be restarted before going on") from e
            while len(self._graphed_fwd) >= max(1, self.max_graphs):
                self._graphed_fwd.pop(next(iter(self._graphed_fwd)))
        else:
            self._graphed_fwd.pop(key)
        self._graphed_fwd[key] = rec          # most recently used last
        return rec.replay(metrics, logs)

    def _forward_impl(self, metrics: torch.Tensor, service_graph: ServiceGraph, logs: Optional[torch.Tensor],
                      return_intermediate: bool = False) -> Dict[str, Any]:
        dev = self.device
        if dev.type != "cuda":
            raise _hip.OceanHipError("OCEANModel runs on an MI355X: move it with .cuda(); there is no CPU path")
        metrics = metrics.to(dev, torch.float32, non_blocking=True)
        batch_size = metrics.size(0)
        graph, node_features = self.device_graph(service_graph)
        if self.overlap_encoders:
            # The two encoders do not depend on each other.  The graph encoder's gather kernels are bound by
            # dependent-load latency and leave the matrix pipes idle, the dilated CNN is bound by the matrix pipes:
            # on separate streams they share the chip.  autograd runs each backward on its forward's stream, so
            # the two backward passes overlap the same way.
            cur = torch.cuda.current_stream(dev)
            side = self._encoder_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                spatial = self.spatial_encoder.forward_graph(node_features, graph, batch_size)
            temporal = self.temporal_encoder(metrics)
            cur.wait_stream(side)
            spatial.record_stream(cur)
        else:
            temporal = self.temporal_encoder(metrics)
            spatial = self.spatial_encoder.forward_graph(node_features, graph, batch_size)

        if logs is None:
            log_features = torch.zeros(batch_size, self.log_input_dim, device=dev)
        else:
            logs = logs.to(dev, torch.float32, non_blocking=True)
            log_features = ops.mean_tokens(logs) if logs.dim() == 3 else logs

        attended = self.attention_module(temporal, spatial, log_features)
        fusion_results = self.fusion_module(attended, spatial, attended, compute_contrastive_loss=self.training)
        fused = fusion_results["fused_representation"]
        p = float(self.prediction_head[2].p)
        logits, probs = _HeadFunction.apply(fused, self.training, p, *self._head_params())

        outputs: Dict[str, Any] = {"root_cause_logits": logits, "root_cause_probs": probs, "prediction": probs}
        if self.training and "contrastive_losses" in fusion_results:
            outputs["contrastive_losses"] = fusion_results["contrastive_losses"]
        if return_intermediate:
            outputs["intermediate"] = {"temporal_features": temporal, "spatial_features": spatial,
                                       "attended_features": attended, "fused_representation": fused,
                                       "fusion_details": fusion_results}
        return outputs

    def compute_loss(self, predictions: Dict[str, Any], targets: torch.Tensor, reduction: str = "mean") -> Dict[str, Any]:
        losses: Dict[str, Any] = ops.LossDict()
        probs = predictions["root_cause_probs"]
        contrastive = predictions.get("contrastive_losses", {})
        stacked = getattr(contrastive, "stacked", None)
        if reduction in ("mean", "sum") and probs.is_cuda and probs.shape == targets.shape and (stacked is not None or not contrastive):
            # the prediction loss and the scalar arithmetic below as one launch (and their backward as one)
            total, pred_loss, csum, vec = ops.prediction_and_total(probs, targets, reduction, stacked, self.prediction_loss_weight,
                                                                   self.contrastive_loss_weight)
            losses["prediction_loss"] = pred_loss
            for name, value in contrastive.items():
                losses[f"contrastive_{name}"] = value
            losses["contrastive_loss"] = csum if contrastive else 0.0
            losses["total_loss"] = total
            losses.vec = vec
            return losses
        pred_loss = ops.binary_cross_entropy(probs, targets, reduction)
        losses["prediction_loss"] = pred_loss
        for name, value in contrastive.items():
            losses[f"contrastive_{name}"] = value
        if pred_loss.dim() == 0 and pred_loss.is_cuda and (stacked is not None or not contrastive):
            # the scalar arithmetic below as one launch (and its backward as one)
            total, csum, vec = ops.combine_losses(pred_loss, stacked, self.prediction_loss_weight, self.contrastive_loss_weight)
            losses["contrastive_loss"] = csum if contrastive else 0.0
            losses["total_loss"] = total
            losses.vec = vec
            return losses
        total_c: Any = 0.0
        for value in contrastive.values():
            total_c = total_c + value
        losses["contrastive_loss"] = total_c
        losses["total_loss"] = self.prediction_loss_weight * pred_loss + self.contrastive_loss_weight * total_c
        return losses

    def predict_root_cause(self, metrics, service_graph: ServiceGraph, logs, threshold: float = 0.5,
                           top_k: Optional[int] = None) -> Dict[str, Any]:
        self.eval()
        with torch.no_grad():
            probs = self.forward(metrics, service_graph, logs)["root_cause_probs"]
            results: Dict[str, Any] = {"probabilities": probs, "binary_predictions": (probs > threshold).float(),
                                       "threshold": threshold}
            if top_k is not None:
                tk = torch.topk(probs, min(top_k, probs.size(-1)), dim=-1)
                results.update({"top_k_probabilities": tk.values, "top_k_indices": tk.indices, "top_k": top_k})
            if service_graph.service_names:
                results["service_names"] = service_graph.service_names
                if top_k is not None:
                    idx = results["top_k_indices"].cpu().tolist()
                    results["top_k_services"] = [[service_graph.service_names[i] for i in row] for row in idx]
            return results

    def get_attention_weights(self, metrics, service_graph: ServiceGraph, logs) -> Dict[str, Any]:
        self.eval()
        with torch.no_grad():
            out = self.forward(metrics, service_graph, logs, return_intermediate=True)
            it = out["intermediate"]
            lg = logs.to(self.device, torch.float32)
            lg = ops.mean_tokens(lg) if lg.dim() == 3 else lg
            return {"attention_analysis": self.attention_module.get_attention_analysis(
                        it["temporal_features"], it["spatial_features"], lg),
                    "fusion_analysis": self.fusion_module.get_fusion_analysis(
                        it["attended_features"], it["spatial_features"], it["attended_features"]),
                    "temporal_features_norm": torch.norm(it["temporal_features"], dim=1),
                    "spatial_features_norm": torch.norm(it["spatial_features"], dim=1),
                    "fused_features_norm": torch.norm(it["fused_representation"], dim=1)}

    def get_model_summary(self) -> Dict[str, Any]:
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return {"model_type": "OCEAN", "total_parameters": total, "trainable_parameters": trainable,
                "parameter_efficiency": trainable / total,
                "model_config": {"temporal_dim": self.temporal_dim, "spatial_dim": self.spatial_dim,
                                 "attention_dim": self.attention_dim, "fusion_dim": self.fusion_dim,
                                 "num_services": self.config.model.num_services,
                                 "prediction_loss_weight": self.prediction_loss_weight,
                                 "contrastive_loss_weight": self.contrastive_loss_weight},
                "component_info": {"temporal_encoder": self.temporal_encoder.get_model_info(),
                                   "spatial_encoder": self.spatial_encoder.get_model_info(),
                                   "attention_module": self.attention_module.get_model_info(),
                                   "fusion_module": self.fusion_module.get_model_info()}}

    def _component(self, name: str) -> nn.Module:
        table = {"temporal": self.temporal_encoder, "spatial": self.spatial_encoder, "attention": self.attention_module,
                 "fusion": self.fusion_module, "prediction": self.prediction_head}
        if name not in table:
            raise ValueError(f"Unknown component: {name}")
        return table[name]

    def freeze_component(self, component_name: str):
        for p in self._component(component_name).parameters():
            p.requires_grad = False
        self.trainable_version = getattr(self, "trainable_version", 0) + 1     # recorded update steps are keyed on it

    def unfreeze_component(self, component_name: str):
        for p in self._component(component_name).parameters():
            p.requires_grad = True
        self.trainable_version = getattr(self, "trainable_version", 0) + 1
