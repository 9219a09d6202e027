"""Graph fusion with contrastive (InfoNCE) learning, on MI355X.

Interface and state_dict keys follow the reference's ProjectionHead,
InfoNCELoss, ContrastiveLearningModule and GraphFusionModule
(ocean/models/components/graph_fusion.py:17-92, 95-162, 165-273, 276-501).
torch sub-modules are parameter containers; the arithmetic runs in
libocean_hip.so (ocean_fusion_fwd / ocean_fusion_bwd, ocean_infonce_*).

Two documented differences from the reference, both where the reference
fails: the memory-bank write is a modulo ring for every batch size (the
reference raises as soon as ptr + 3B > 2*16, graph_fusion.py:263-273), and
`negatives='in_batch'` selects InfoNCELoss's in-batch branch
(graph_fusion.py:144-157), which the reference model never reaches.
"""
from __future__ import annotations

import os

import ctypes
from typing import Any, Callable, Dict, Optional

import torch
import torch.nn as nn

from ... import _hip, ops


class ProjectionHead(nn.Module):
    """Linear -> ReLU -> Dropout -> Linear -> L2 normalise (graph_fusion.py:23-92)."""

    def __init__(self, input_dim: int, projection_dim: int = 256, hidden_dim: Optional[int] = None, num_layers: int = 2,
                 activation: str = "relu", dropout: float = 0.1):
        super().__init__()
        hidden_dim = input_dim if hidden_dim is None else hidden_dim
        layers, cur = [], input_dim
        for i in range(num_layers):
            if i == num_layers - 1:
                layers.append(nn.Linear(cur, projection_dim))
            else:
                layers.append(nn.Linear(cur, hidden_dim))
                layers.append({"relu": nn.ReLU, "gelu": nn.GELU, "leaky_relu": nn.LeakyReLU}.get(activation, nn.ReLU)())
                if dropout > 0:
                    layers.append(nn.Dropout(dropout))
                cur = hidden_dim
        self.projection = nn.Sequential(*layers)
        self.normalize = True
        self.num_layers, self.activation_name = num_layers, activation

    def _two_layer_params(self):
        lin = [m for m in self.projection if isinstance(m, nn.Linear)]
        drop = [m for m in self.projection if isinstance(m, nn.Dropout)]
        if self.num_layers != 2 or self.activation_name != "relu" or len(lin) != 2:
            raise NotImplementedError("the HIP path implements the reference's default projection head: 2 layers with ReLU")
        return [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias], (drop[0].p if drop else 0.0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The head on its own (graph_fusion.py:77-92), as ContrastiveLearningModule.forward calls it per modality;
        GraphFusionModule runs its three heads inside its fused driver instead."""
        params, p = self._two_layer_params()
        if x.dim() != 2 or x.shape[1] != params[0].shape[1]:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} and "
                               f"{params[0].shape[1]}x{params[0].shape[0]})")
        return _ProjectionFunction.apply(x.contiguous().float(), self, float(p), *params)


class _ProjectionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, p, *params):
        L = _hip.lib()
        sh = _hip.ProjShape()
        sh.B, sh.in_dim, sh.hidden, sh.out_dim = x.shape[0], x.shape[1], params[0].shape[0], params[2].shape[0]
        sh.training, sh.normalize = int(module.training), int(bool(module.normalize))
        sh.dropout = p if module.training else 0.0
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(x.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        ws = torch.empty(L.ocean_projection_head_fwd_workspace(ctypes.byref(sh)), device=x.device, dtype=torch.float32)
        y = torch.empty(sh.B, sh.out_dim, device=x.device, dtype=torch.float32)
        _hip.check(L.ocean_projection_head_fwd(ctypes.byref(sh), _hip.dev_ptr(x, "x"), _hip.ptr_table(params, "projection param"),
                                               _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(y), _hip.stream()),
                   "ocean_projection_head_fwd")
        ctx.sh = sh
        ctx.save_for_backward(x, ws, y, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * (3 + 4)
        L = _hip.lib()
        x, ws, y, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_projection_head_bwd_workspace(ctypes.byref(sh)), device=x.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _hip.check(L.ocean_projection_head_bwd(ctypes.byref(sh), _hip.dev_ptr(x), _hip.ptr_table(params, "projection param"),
                                               _hip.dev_ptr(ws), _hip.dev_ptr(y), _hip.dev_ptr(dy.contiguous()), _hip.dev_ptr(wb),
                                               wb.numel(), _hip.ptr_table(grads, "projection grad"), _hip.dev_ptr(dx),
                                               _hip.stream()), "ocean_projection_head_bwd")
        return (dx, None, None) + tuple(g if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads))


class InfoNCELoss(nn.Module):
    def __init__(self, temperature: float = 0.1, reduction: str = "mean"):
        super().__init__()
        self.temperature, self.reduction = temperature, reduction

    def forward(self, anchor: torch.Tensor, positive: torch.Tensor, negative: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.reduction not in ("mean", "sum", "none"):
            raise ValueError(f"{self.reduction} is not a valid value for reduction")
        if self.reduction == "none":              # F.cross_entropy(..., reduction='none'): one value per anchor row
            return ops.infonce(anchor, positive, negative, self.temperature, per_row=True)
        loss = ops.infonce(anchor, positive, negative, self.temperature)
        return loss * anchor.shape[0] if self.reduction == "sum" else loss


def _no_row_offset(local_rows: int) -> int:
    """A single process holds the whole batch: its rows start at 0 (a module-level function, so the module pickles)."""
    return 0


class ContrastiveLearningModule(nn.Module):
    def __init__(self, modality_dims: Dict[str, int], projection_dim: int = 256, temperature: float = 0.1,
                 num_negatives: int = 16):
        super().__init__()
        self.modality_dims, self.projection_dim, self.num_negatives = modality_dims, projection_dim, num_negatives
        self.projection_heads = nn.ModuleDict({m: ProjectionHead(d, projection_dim) for m, d in modality_dims.items()})
        self.infonce_loss = InfoNCELoss(temperature)
        self.register_buffer("memory_bank", torch.randn(num_negatives, projection_dim))
        self.register_buffer("memory_ptr", torch.zeros(1, dtype=torch.long))
        self.negatives = "bank"                    # or "in_batch"
        # data-parallel hooks (in-batch negatives): gather [B_loc,P] -> [B_glob,P] keeping autograd, and this rank's row offset
        self.gather_positives: Optional[Callable[[torch.Tensor], torch.Tensor]] = None
        self.rank_row_offset: Callable[[int], int] = _no_row_offset
        # data-parallel hook (memory-bank negatives): [M, B_loc, P] -> [M, B_all, P], the projections of every rank in
        # rank order, so that each rank writes the whole batch's rows into its copy of the ring and the copies stay equal
        self.gather_bank_rows: Optional[Callable[[torch.Tensor], torch.Tensor]] = None
        self.device_ring = True                    # the ring pointer lives in memory_ptr on the device (capturable)
        self.batch_pairs = os.environ.get("OCEAN_INFONCE_PER_PAIR") != "1"     # measurement switch: one call per pair

    def _get_negatives(self) -> torch.Tensor:
        return self.memory_bank.clone().detach()

    def losses_from_projections(self, projections: Dict[str, torch.Tensor],
                                stacked: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """InfoNCE over every pair of modalities (graph_fusion.py:231-249).  `stacked` ([3, B, P], the same three
        projections in dictionary order, as the fused driver produces them) lets the single-process in-batch case run
        the three pairs as one batched call."""
        names = list(projections.keys())
        losses = {}
        neg = self._get_negatives() if self.negatives == "bank" else None
        if neg is None and stacked is not None and len(names) == 3 and stacked.shape[0] == 3 and self.batch_pairs:
            gathered, offset = None, 0
            if self.gather_positives is not None:
                # data parallel: one collective over the three projections side by side (its backward: one reduce-scatter)
                B, P = stacked.shape[1], stacked.shape[2]
                allz = self.gather_positives(stacked.permute(1, 0, 2).reshape(B, 3 * P))          # [B_all, 3P]
                gathered, offset = allz.view(-1, 3, P).permute(1, 0, 2), self.rank_row_offset(B)
            vec = ops.infonce_pairs(stacked, self.infonce_loss.temperature, gathered, offset)      # one backward node
            three = vec.unbind(0)
            pairs = [(a, b) for i, a in enumerate(names) for b in names[i + 1:]]
            out = ops.StackedLosses({f"{a}_{b}_contrastive": three[q] for q, (a, b) in enumerate(pairs)})
            out.stacked = vec          # the same values as one tensor, in the dictionary's order (OCEANModel.compute_loss)
            return out
        gathered = {}
        if neg is None and self.gather_positives is not None and len(names) > 1:
            # data parallel: every modality but the first serves as the positive side of some pair; one collective
            # over their column-wise concatenation (and one reduce-scatter in backward) instead of one per pair
            pos = names[1:]
            P = projections[pos[0]].shape[1]
            allp = self.gather_positives(torch.cat([projections[n] for n in pos], 1))
            gathered = {n: allp[:, k * P:(k + 1) * P].contiguous() for k, n in enumerate(pos)}
        for i, a in enumerate(names):
            for b in names[i + 1:]:
                pa, pb = projections[a], projections[b]
                if neg is not None:
                    losses[f"{a}_{b}_contrastive"] = ops.infonce(pa, pb, neg, self.infonce_loss.temperature)
                else:
                    allb = gathered.get(b, pb)
                    losses[f"{a}_{b}_contrastive"] = ops.infonce(pa, allb, None, self.infonce_loss.temperature,
                                                                  label_offset=self.rank_row_offset(pa.shape[0]))
        return losses

    def _update_memory_bank(self, projections, stacked: Optional[torch.Tensor] = None) -> None:
        """Ring write of cat(projections) at memory_ptr.  Contents equal the reference's on every step it survives; the
        stored pointer follows its formula (including the out-of-range value K when ptr + R == 2K).  `stacked`
        [M, B, P] holds the same projections in dictionary order (as the fused driver produces them): no concatenation.
        Data parallel: the rows of all ranks are gathered first, so every rank performs the identical write."""
        if not projections:
            return
        if stacked is None:
            stacked = torch.stack([p.detach() for p in projections.values()], dim=0)
        stacked = stacked.detach()
        if self.gather_bank_rows is not None:
            stacked = self.gather_bank_rows(stacked)
        rows = stacked.reshape(-1, stacked.shape[-1]).to(self.memory_bank.dtype).contiguous()
        K, P = self.memory_bank.shape
        if rows.shape[1] != P:
            raise ValueError(f"projection width {rows.shape[1]} does not match the memory bank's {P}")
        _hip.check(_hip.lib().ocean_bank_ring_write(_hip.dev_ptr(self.memory_bank, "memory_bank"),
                                                    _hip.dev_ptr(self.memory_ptr, "memory_ptr", torch.int64),
                                                    _hip.dev_ptr(rows, "rows"), rows.shape[0], K, P, _hip.stream()),
                   "ocean_bank_ring_write")

    def forward(self, modality_features: Dict[str, torch.Tensor], return_projections: bool = False) -> Dict[str, Any]:
        """The module on its own (graph_fusion.py:204-245): every modality that has a head is projected, InfoNCE over
        every pair of modalities (against the memory bank, or in-batch when `negatives` says so), then - in training -
        the ring write.  GraphFusionModule does the same from the projections its fused driver produces."""
        projections = {m: self.projection_heads[m](x) for m, x in modality_features.items() if m in self.projection_heads}
        losses = self.losses_from_projections(projections)
        if self.training and self.negatives == "bank":
            self._update_memory_bank(projections)
        results: Dict[str, Any] = {"losses": dict(losses)}
        if return_projections:
            results["projections"] = projections
        return results


class _FusionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, s, l, module, contrastive, *params):
        L = _hip.lib()
        sh = module._shape(t, s, l, contrastive)
        ctx.set_materialize_grads(False)
        dev = t.device
        ws = torch.empty(L.ocean_fusion_fwd_workspace(ctypes.byref(sh)), device=dev, dtype=torch.float32)
        fused = torch.empty(sh.B, sh.E, device=dev, dtype=torch.float32)
        X = torch.empty(sh.B, 3, sh.E, device=dev, dtype=torch.float32)
        Zp = torch.empty(3, sh.B, sh.Pd, device=dev, dtype=torch.float32) if contrastive else None
        _hip.check(L.ocean_fusion_fwd(ctypes.byref(sh), _hip.dev_ptr(t, "temporal"), _hip.dev_ptr(s, "spatial"), _hip.dev_ptr(l, "logs"),
                                      _hip.ptr_table(params, "fusion param"), _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(fused),
                                      _hip.dev_ptr(X), _hip.dev_ptr(Zp), _hip.stream()), "ocean_fusion_fwd")
        ctx.sh, ctx.contrastive = sh, contrastive
        ctx.save_for_backward(t, s, l, ws, X, Zp if contrastive else X, *params)
        if contrastive:
            return fused, X, Zp
        return fused, X, torch.empty(0, device=dev)

    @staticmethod
    def backward(ctx, dfused, dX, dZp):
        L = _hip.lib()
        t, s, l, ws, X, Zp, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_fusion_bwd_workspace(ctypes.byref(sh)), device=t.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        dt = torch.empty_like(t) if ctx.needs_input_grad[0] else None
        ds = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        dl = torch.empty_like(l) if ctx.needs_input_grad[2] else None
        c = lambda g: None if g is None else g.contiguous()
        if not ctx.contrastive:
            dZp = None
        _hip.check(L.ocean_fusion_bwd(ctypes.byref(sh), _hip.dev_ptr(t), _hip.dev_ptr(s), _hip.dev_ptr(l),
                                      _hip.ptr_table(params, "fusion param"), _hip.dev_ptr(ws), _hip.dev_ptr(X),
                                      _hip.dev_ptr(Zp if ctx.contrastive else None), _hip.dev_ptr(wb), wb.numel(),
                                      _hip.dev_ptr(c(dfused)), _hip.dev_ptr(c(dX)), _hip.dev_ptr(c(dZp)),
                                      _hip.ptr_table(grads, "fusion grad"), _hip.dev_ptr(dt), _hip.dev_ptr(ds), _hip.dev_ptr(dl),
                                      _hip.stream()), "ocean_fusion_bwd")
        n_head = 12
        out = [dt, ds, dl, None, None]
        for i, g in enumerate(grads):
            used = ctx.contrastive or i < 22       # projection-head parameters only receive gradients when they ran
            out.append(g if (used and ctx.needs_input_grad[5 + i]) else None)
        return tuple(out)


class GraphFusionModule(nn.Module):
    def __init__(self, temporal_dim: int, spatial_dim: int, log_dim: int, fusion_dim: int = 256, projection_dim: int = 128,
                 temperature: float = 0.1, fusion_strategy: str = "attention", dropout: float = 0.1):
        super().__init__()
        self.temporal_dim, self.spatial_dim, self.log_dim = temporal_dim, spatial_dim, log_dim
        self.fusion_dim, self.projection_dim, self.fusion_strategy = fusion_dim, projection_dim, fusion_strategy
        mk = lambda d: nn.Sequential(nn.Linear(d, fusion_dim), nn.ReLU(), nn.Dropout(dropout), nn.LayerNorm(fusion_dim))
        self.temporal_processor, self.spatial_processor, self.log_processor = mk(temporal_dim), mk(spatial_dim), mk(log_dim)
        self.contrastive_module = ContrastiveLearningModule(
            {"temporal": fusion_dim, "spatial": fusion_dim, "log": fusion_dim}, projection_dim, temperature)
        if fusion_strategy == "attention":
            self.attention_fusion = nn.MultiheadAttention(embed_dim=fusion_dim, num_heads=8, dropout=dropout, batch_first=True)
            self.attention_norm = nn.LayerNorm(fusion_dim)
        elif fusion_strategy == "concatenate":
            self.fusion_layer = nn.Sequential(nn.Linear(fusion_dim * 3, fusion_dim), nn.ReLU(), nn.Dropout(dropout),
                                              nn.LayerNorm(fusion_dim))
        elif fusion_strategy == "gate":
            self.gate_network = nn.Sequential(nn.Linear(fusion_dim * 3, fusion_dim), nn.Sigmoid())
            self.value_network = nn.Sequential(nn.Linear(fusion_dim * 3, fusion_dim), nn.Tanh())
        self.output_projection = nn.Sequential(nn.Linear(fusion_dim, fusion_dim), nn.ReLU(), nn.Dropout(dropout),
                                               nn.Linear(fusion_dim, fusion_dim))
        self.mi_estimator = nn.Sequential(nn.Linear(fusion_dim * 2, fusion_dim), nn.ReLU(), nn.Linear(fusion_dim, 1))
        self.compute_mutual_information = True

    def _param_list(self):
        ps = []
        for proc in (self.temporal_processor, self.spatial_processor, self.log_processor):
            ps += [proc[0].weight, proc[0].bias, proc[3].weight, proc[3].bias]
        af, op = self.attention_fusion, self.output_projection
        ps += [af.in_proj_weight, af.in_proj_bias, af.out_proj.weight, af.out_proj.bias,
               self.attention_norm.weight, self.attention_norm.bias, op[0].weight, op[0].bias, op[3].weight, op[3].bias]
        for m in ("temporal", "spatial", "log"):
            pr = self.contrastive_module.projection_heads[m].projection
            ps += [pr[0].weight, pr[0].bias, pr[3].weight, pr[3].bias]
        return ps

    def _shape(self, t, s, l, contrastive) -> _hip.FusionShape:
        sh = _hip.FusionShape()
        sh.B, sh.td, sh.sd, sh.ld = t.shape[0], t.shape[1], s.shape[1], l.shape[1]
        sh.E, sh.Pd, sh.heads = self.fusion_dim, self.projection_dim, self.attention_fusion.num_heads
        sh.training, sh.contrastive = int(self.training), int(contrastive)
        sh.dropout = float(self.output_projection[2].p) if self.training else 0.0
        sh.eps = self.attention_norm.eps
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(t.device).data_ptr() if sh.dropout > 0 else None
        return sh

    def forward(self, temporal_features: torch.Tensor, spatial_features: torch.Tensor, log_features: torch.Tensor,
                compute_contrastive_loss: bool = True) -> Dict[str, Any]:
        if self.fusion_strategy != "attention":
            raise NotImplementedError("the HIP path implements fusion_strategy='attention' (the one OCEANModel uses)")
        ph = self.contrastive_module.projection_heads["temporal"]
        if ph.num_layers != 2 or ph.activation_name != "relu":
            raise NotImplementedError("projection heads: 2 layers with ReLU")
        t, s, l = (x.contiguous().float() for x in (temporal_features, spatial_features, log_features))
        fused, X, Zp = _FusionFunction.apply(t, s, l, self, bool(compute_contrastive_loss), *self._param_list())
        processed = {"temporal": X[:, 0], "spatial": X[:, 1], "log": X[:, 2]}
        results: Dict[str, Any] = {"fused_representation": fused, "processed_features": processed}
        if compute_contrastive_loss:
            cm = self.contrastive_module
            projections = {"temporal": Zp[0], "spatial": Zp[1], "log": Zp[2]}
            results["contrastive_losses"] = cm.losses_from_projections(projections, stacked=Zp)
            if cm.training and cm.negatives == "bank":
                cm._update_memory_bank(projections, stacked=Zp)
            results["projections"] = projections
            results["mutual_information"] = self._mutual_information(fused, X) if self.compute_mutual_information else {}
        return results

    @staticmethod
    def draw_permutations(B: int, device) -> torch.Tensor:
        """Three independent uniform row permutations [3, B] (the reference draws torch.randperm(B) once per modality,
        graph_fusion.py:494), from one launch: the sorted order of counter-based random keys, seeded from torch's CPU
        generator (plus the device epoch counter, so that a replayed HIP graph draws new ones each time)."""
        device = torch.device(device)
        if device.type == "cuda" and B <= 4096:
            out = torch.empty(3, B, device=device, dtype=torch.long)
            with torch.cuda.device(device):
                _hip.check(_hip.lib().ocean_draw_permutations(3, B, _hip.new_seed(1.0), _hip.seed_epoch(device).data_ptr(),
                                                              out.data_ptr(), _hip.stream()), "ocean_draw_permutations")
            return out
        return torch.rand(3, B, device=device).argsort(dim=1)

    def _mutual_information(self, fused, X, perms: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """MINE estimates of graph_fusion.py:475-501 (diagnostics only: they enter no loss).  X is [B, 3, E], the
        processed temporal / spatial / log features; perms [3, B] are the shuffles of the marginal term."""
        w0, b0, w1, b1 = (p.detach().contiguous() for p in (self.mi_estimator[0].weight, self.mi_estimator[0].bias,
                                                            self.mi_estimator[2].weight, self.mi_estimator[2].bias))
        B, E = fused.shape
        with torch.no_grad():
            dev = fused.device
            if perms is None:
                perms = self.draw_permutations(B, dev)
            perms = perms.to(device=dev, dtype=torch.long).contiguous()
            L = _hip.lib()
            sh = _hip.MiShape()
            sh.B, sh.E = B, E
            ws = torch.empty(L.ocean_mi_workspace(ctypes.byref(sh)), device=dev, dtype=torch.float32)
            mi = torch.empty(3, device=dev, dtype=torch.float32)
            _hip.check(L.ocean_mi_estimates(ctypes.byref(sh), _hip.dev_ptr(fused.detach().contiguous(), "fused"),
                                            _hip.dev_ptr(X.detach().contiguous(), "processed"), ctypes.c_void_p(perms.data_ptr()),
                                            _hip.dev_ptr(w0), _hip.dev_ptr(b0), _hip.dev_ptr(w1), _hip.dev_ptr(b1),
                                            _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(mi), _hip.stream()), "ocean_mi_estimates")
        return {f"{k}_mi": mi[i] for i, k in enumerate(("temporal", "spatial", "log"))}

    def get_fusion_analysis(self, temporal_features, spatial_features, log_features) -> Dict[str, Any]:
        with torch.no_grad():
            r = self.forward(temporal_features, spatial_features, log_features)
        a = {"modality_contributions": {k: torch.norm(v, dim=1).mean().item() for k, v in r["processed_features"].items()},
             "fusion_strategy": self.fusion_strategy}
        if "mutual_information" in r:
            a["mutual_information"] = {k: v.item() for k, v in r["mutual_information"].items()}
        if "contrastive_losses" in r:
            a["contrastive_losses"] = {k: v.item() for k, v in r["contrastive_losses"].items()}
        return a

    def get_model_info(self) -> Dict[str, Any]:
        total = sum(p.numel() for p in self.parameters())
        return {"model_type": "GraphFusionModule", "temporal_dim": self.temporal_dim, "spatial_dim": self.spatial_dim,
                "log_dim": self.log_dim, "fusion_dim": self.fusion_dim, "fusion_strategy": self.fusion_strategy,
                "total_parameters": total,
                "trainable_parameters": sum(p.numel() for p in self.parameters() if p.requires_grad)}
