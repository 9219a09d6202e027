"""Multi-factor attention over the temporal / spatial / log modality vectors, on MI355X.

Interface and state_dict keys follow the reference's MultiFactorAttention and
the helper classes it instantiates (ocean/models/components/
multi_factor_attention.py:17-215, 218-440).  The helper classes here are
parameter containers (the reference constructs three CrossModalAttention
blocks it never calls, :264-266 - their parameters must still exist in the
state_dict); the arithmetic of MultiFactorAttention.forward runs in
libocean_hip.so (ocean_mfa_fwd / ocean_mfa_bwd).
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict

import torch
import torch.nn as nn

from ... import _hip


class ScaledDotProductAttention(nn.Module):
    def __init__(self, d_model: int, dropout: float = 0.1):
        super().__init__()
        self.d_model = d_model
        self.dropout = nn.Dropout(dropout)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, mask=None):
        """The attention alone (multi_factor_attention.py:34-65): softmax(Q K^T / sqrt(d_model)) V with dropout on the
        weights, for [batch, seq_len, d] or [batch, heads, seq_len, d] inputs; mask broadcastable to the scores,
        0 = masked.  Returns (attended values, attention weights) in the inputs' layout."""
        if query.dim() not in (3, 4) or key.shape != query.shape or value.shape != query.shape:
            raise RuntimeError(f"query, key and value of one shape [batch, (heads,) seq_len, d] expected, got "
                               f"{tuple(query.shape)}, {tuple(key.shape)}, {tuple(value.shape)}")
        four = query.dim() == 4
        B, H = query.shape[0], (query.shape[1] if four else 1)
        L, d = query.shape[-2], query.shape[-1]
        rows = lambda x: (x.permute(0, 2, 1, 3) if four else x).contiguous().float().view(B, L, H * d)
        m = None
        if mask is not None:
            m = (torch.as_tensor(mask, device=query.device) != 0).to(torch.float32)
            if m.dim() == 4:
                if m.shape[1] != 1 and H > 1 and not bool((m == m[:, :1]).all()):
                    raise NotImplementedError("one mask for all heads of a sample")
                m = m[:, 0]
            m = m.expand(B, L, L).contiguous()
        out, w = _MhaFunction.apply(rows(query), rows(key), rows(value), m, self, H, 1.0 / (self.d_model ** 0.5))
        if four:
            return out.view(B, L, H, d).permute(0, 2, 1, 3).contiguous(), w
        return out, w[:, 0]


class _MhaFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, module, *params):
        # params: the eight tensors of MultiHeadAttention, or (heads, scale) for the attention alone
        L = _hip.lib()
        sh = _hip.MhaShape()
        project = len(params) == 8
        sh.B, sh.L, sh.D = q.shape[0], q.shape[1], q.shape[2]
        sh.heads, sh.scale = (module.num_heads, 0.0) if project else (int(params[0]), float(params[1]))
        sh.project = int(project)
        if not project:
            params = ()
        sh.training = int(module.training)
        sh.dropout = float(module.dropout.p) if module.training else 0.0
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(q.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        nws = L.ocean_mha_fwd_workspace(ctypes.byref(sh))
        if nws == 0:
            raise _hip.OceanHipError(f"MultiHeadAttention: unsupported shape {tuple(q.shape)} with {module.num_heads} heads "
                                     "(sequence length at most 1024)")
        ws = torch.empty(nws, device=q.device, dtype=torch.float32)
        out = torch.empty_like(q)
        attn = torch.empty(sh.B, sh.heads, sh.L, sh.L, device=q.device, dtype=torch.float32)
        _hip.check(L.ocean_mha_fwd(ctypes.byref(sh), _hip.dev_ptr(q, "query"), _hip.dev_ptr(k, "key"), _hip.dev_ptr(v, "value"),
                                   _hip.dev_ptr(mask, "mask"), _hip.ptr_table(params, "attention param") if project else None,
                                   _hip.dev_ptr(ws),
                                   ws.numel(), _hip.dev_ptr(out), _hip.dev_ptr(attn), _hip.stream()), "ocean_mha_fwd")
        ctx.sh = sh
        ctx.save_for_backward(q, k, v, ws, attn, *[p for p in params if p is not None])
        ctx.has = [p is not None for p in params]
        return out, attn

    @staticmethod
    def backward(ctx, d_out, d_attn):
        project = bool(ctx.sh.project)
        n_in = 5 + (len(ctx.has) if project else 2)
        if d_out is None and d_attn is None:
            return (None,) * n_in
        L = _hip.lib()
        q, k, v, ws, attn, *live = ctx.saved_tensors
        it = iter(live)
        params = [next(it) if h else None for h in ctx.has]
        sh = ctx.sh
        wb = torch.empty(L.ocean_mha_bwd_workspace(ctypes.byref(sh)), device=q.device, dtype=torch.float32)
        present = [p for p in params if p is not None]
        got = iter(_hip.zero_grads_like(present) if present else [])
        grads = [next(got) if p is not None else None for p in params]
        dq = torch.empty_like(q) if (ctx.needs_input_grad[0] or not project) else None
        dk = torch.empty_like(k) if (ctx.needs_input_grad[1] or not project) else None
        dv = torch.empty_like(v) if (ctx.needs_input_grad[2] or not project) else None
        c = lambda g: None if g is None else g.contiguous().float()
        _hip.check(L.ocean_mha_bwd(ctypes.byref(sh), _hip.dev_ptr(q), _hip.dev_ptr(k), _hip.dev_ptr(v),
                                   _hip.ptr_table(params, "attention param") if project else None, _hip.dev_ptr(ws),
                                   _hip.dev_ptr(attn), _hip.dev_ptr(c(d_out)), _hip.dev_ptr(c(d_attn)), _hip.dev_ptr(wb), wb.numel(),
                                   _hip.ptr_table(grads, "attention grad") if project else None, _hip.dev_ptr(dq), _hip.dev_ptr(dk),
                                   _hip.dev_ptr(dv), _hip.stream()), "ocean_mha_bwd")
        if not project:
            return (dq, dk, dv, None, None, None, None)
        return (dq, dk, dv, None, None) + tuple(g if (g is not None and ctx.needs_input_grad[5 + i]) else None
                                                 for i, g in enumerate(grads))


class MultiHeadAttention(nn.Module):
    """w_q / w_k / w_v / w_o of multi_factor_attention.py:96-101.  On the model
    path it only ever sees a length-1 sequence, where softmax == 1 and w_q / w_k
    cannot influence the output."""

    def __init__(self, d_model: int, num_heads: int, dropout: float = 0.1, bias: bool = True):
        super().__init__()
        assert d_model % num_heads == 0, "d_model must be divisible by num_heads"
        self.d_model, self.num_heads, self.d_k = d_model, num_heads, d_model // num_heads
        self.w_q = nn.Linear(d_model, d_model, bias=bias)
        self.w_k = nn.Linear(d_model, d_model, bias=bias)
        self.w_v = nn.Linear(d_model, d_model, bias=bias)
        self.w_o = nn.Linear(d_model, d_model, bias=bias)
        self.attention = ScaledDotProductAttention(self.d_k, dropout)
        self.dropout = nn.Dropout(dropout)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, mask=None):
        """The module on its own (multi_factor_attention.py:108-148): (output [B, L, D], attention weights [B, heads, L, L]
        after dropout).  As in the reference, key and value are viewed with the query's sequence length, so the three
        must agree in shape; mask [B, L, L] (or broadcastable to it), 0 = masked."""
        if query.dim() != 3 or query.shape[2] != self.d_model:
            raise RuntimeError(f"expected query of shape [batch, seq_len, {self.d_model}], got {tuple(query.shape)}")
        if key.shape != query.shape or value.shape != query.shape:
            raise RuntimeError(f"shape '[{query.shape[0]}, {query.shape[1]}, {self.num_heads}, {self.d_k}]' is invalid for input "
                               f"of size {key.numel() if key.shape != query.shape else value.numel()}")
        if self.attention.dropout.p != self.dropout.p:
            raise NotImplementedError("one dropout probability for the attention weights and the output")
        B, L, _ = query.shape
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=query.device)
            m = (m != 0).to(torch.float32).expand(B, L, L).contiguous()
        q, k, v = (x.contiguous().float() for x in (query, key, value))
        params = [self.w_q.weight, self.w_q.bias, self.w_k.weight, self.w_k.bias, self.w_v.weight, self.w_v.bias,
                  self.w_o.weight, self.w_o.bias]
        return _MhaFunction.apply(q, k, v, m, self, *params)


class CrossModalAttention(nn.Module):
    """Constructed but never called by the reference model (multi_factor_attention.py:264-266)."""

    def __init__(self, d_model: int, num_heads: int = 8, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.num_heads = d_model, num_heads
        self.multihead_attention = MultiHeadAttention(d_model, num_heads, dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        self.feed_forward = nn.Sequential(nn.Linear(d_model, d_model * 4), nn.ReLU(), nn.Dropout(dropout),
                                          nn.Linear(d_model * 4, d_model), nn.Dropout(dropout))
        self.ff_layer_norm = nn.LayerNorm(d_model)

    def forward(self, query_modality: torch.Tensor, key_value_modality: torch.Tensor, mask=None):
        """The block on its own (multi_factor_attention.py:190-215): attention of the query modality over the other one,
        residual + LayerNorm, feed-forward, residual + LayerNorm.  Returns (output, attention weights)."""
        attended, weights = self.multihead_attention(query_modality, key_value_modality, key_value_modality, mask)
        ff = self.feed_forward
        drops = {m.p for m in ff if isinstance(m, nn.Dropout)}
        if len(drops) > 1 or self.layer_norm.eps != self.ff_layer_norm.eps:
            raise NotImplementedError("one dropout probability and one epsilon per block")
        params = [self.layer_norm.weight, self.layer_norm.bias, ff[0].weight, ff[0].bias, ff[3].weight, ff[3].bias,
                  self.ff_layer_norm.weight, self.ff_layer_norm.bias]
        q = query_modality.contiguous().float()
        out = _XattnTailFunction.apply(q, attended, self, float(next(iter(drops))) if drops else 0.0, *params)
        return out, weights


class _XattnTailFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, attended, module, p, *params):
        L = _hip.lib()
        sh = _hip.XattnTailShape()
        sh.rows, sh.D, sh.F = q.shape[0] * q.shape[1], q.shape[2], params[2].shape[0]
        sh.training, sh.eps = int(module.training), float(module.layer_norm.eps)
        sh.dropout = p if module.training else 0.0
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(q.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        nws = L.ocean_xattn_tail_fwd_workspace(ctypes.byref(sh))
        if nws == 0:
            raise _hip.OceanHipError(f"CrossModalAttention: unsupported width {sh.D} (at most 1024)")
        ws = torch.empty(nws, device=q.device, dtype=torch.float32)
        out = torch.empty_like(q)
        _hip.check(L.ocean_xattn_tail_fwd(ctypes.byref(sh), _hip.dev_ptr(q, "query"), _hip.dev_ptr(attended.contiguous(), "attended"),
                                          _hip.ptr_table(params, "block param"), _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(out),
                                          _hip.stream()), "ocean_xattn_tail_fwd")
        ctx.sh = sh
        ctx.save_for_backward(ws, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if d_out is None:
            return (None,) * (4 + 8)
        L = _hip.lib()
        ws, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_xattn_tail_bwd_workspace(ctypes.byref(sh)), device=ws.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        d_sum = torch.empty_like(d_out, memory_format=torch.contiguous_format)
        _hip.check(L.ocean_xattn_tail_bwd(ctypes.byref(sh), _hip.ptr_table(params, "block param"), _hip.dev_ptr(ws),
                                          _hip.dev_ptr(d_out.contiguous().float()), _hip.dev_ptr(wb), wb.numel(),
                                          _hip.ptr_table(grads, "block grad"), _hip.dev_ptr(d_sum), _hip.stream()),
                   "ocean_xattn_tail_bwd")
        return (d_sum if ctx.needs_input_grad[0] else None, d_sum if ctx.needs_input_grad[1] else None, None, None) + tuple(
            g if ctx.needs_input_grad[4 + i] else None for i, g in enumerate(grads))


class _MfaFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, s, l, module, n_live, *params):
        # params[:n_live] feed the kernels; params[n_live:] are the w_q / w_k tensors of the three
        # length-1 self-attentions: they cannot influence the output and get exact-zero gradients
        # (the reference's autograd gives them zeros too, so its Adam still applies weight decay)
        L = _hip.lib()
        ctx.dead = params[n_live:]
        params = params[:n_live]
        sh = module._shape(t, s, l)
        ws = torch.empty(L.ocean_mfa_fwd_workspace(ctypes.byref(sh)), device=t.device, dtype=torch.float32)
        out = torch.empty(sh.B, sh.W, device=t.device, dtype=torch.float32)
        _hip.check(L.ocean_mfa_fwd(ctypes.byref(sh), _hip.dev_ptr(t, "temporal"), _hip.dev_ptr(s, "spatial"),
                                   _hip.dev_ptr(l, "logs"), _hip.ptr_table(params, "mfa param"), _hip.dev_ptr(ws), ws.numel(),
                                   _hip.dev_ptr(out), _hip.stream()), "ocean_mfa_fwd")
        ctx.sh = sh
        ctx.save_for_backward(t, s, l, ws, *params)
        module._last_ws = ws if module._keep_ws else None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _hip.lib()
        t, s, l, ws, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_mfa_bwd_workspace(ctypes.byref(sh)), device=t.device, dtype=torch.float32)
        # one zero-filled flat buffer carries the live gradients and the dead tensors' zeros (a single fill)
        both = _hip.zero_grads_like(list(params) + list(ctx.dead))
        grads, dead_zero = both[:len(params)], both[len(params):]
        dt = torch.empty_like(t) if ctx.needs_input_grad[0] else None
        ds = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        dl = torch.empty_like(l) if ctx.needs_input_grad[2] else None
        _hip.check(L.ocean_mfa_bwd(ctypes.byref(sh), _hip.dev_ptr(t), _hip.dev_ptr(s), _hip.dev_ptr(l),
                                   _hip.ptr_table(params, "mfa param"), _hip.dev_ptr(ws), _hip.dev_ptr(wb), wb.numel(),
                                   _hip.dev_ptr(dout.contiguous(), "grad_output"), _hip.ptr_table(grads, "mfa grad"),
                                   _hip.dev_ptr(dt), _hip.dev_ptr(ds), _hip.dev_ptr(dl), _hip.stream()), "ocean_mfa_bwd")
        live = tuple(g if ctx.needs_input_grad[5 + i] else None for i, g in enumerate(grads))
        dead = tuple(z if ctx.needs_input_grad[5 + len(grads) + i] else None for i, z in enumerate(dead_zero))
        return (dt, ds, dl, None, None) + live + dead


class MultiFactorAttention(nn.Module):
    def __init__(self, temporal_dim: int, spatial_dim: int, log_dim: int, attention_dim: int = 128, num_heads: int = 8,
                 dropout: float = 0.1, fusion_strategy: str = "concatenate"):
        super().__init__()
        self.temporal_dim, self.spatial_dim, self.log_dim = temporal_dim, spatial_dim, log_dim
        self.attention_dim, self.num_heads, self.fusion_strategy = attention_dim, num_heads, fusion_strategy
        d = attention_dim
        self.temporal_projection = nn.Linear(temporal_dim, d)
        self.spatial_projection = nn.Linear(spatial_dim, d)
        self.log_projection = nn.Linear(log_dim, d)
        self.temporal_self_attention = MultiHeadAttention(d, num_heads, dropout)
        self.spatial_self_attention = MultiHeadAttention(d, num_heads, dropout)
        self.log_self_attention = MultiHeadAttention(d, num_heads, dropout)
        self.temporal_spatial_attention = CrossModalAttention(d, num_heads, dropout)
        self.temporal_log_attention = CrossModalAttention(d, num_heads, dropout)
        self.spatial_log_attention = CrossModalAttention(d, num_heads, dropout)
        self.modality_weights = nn.Parameter(torch.ones(3))
        self.feature_attention = nn.Sequential(nn.Linear(d, d // 2), nn.ReLU(), nn.Linear(d // 2, 1), nn.Sigmoid())
        if fusion_strategy == "concatenate":
            self.fusion_dim = d * 3
            self.fusion_layer = nn.Linear(self.fusion_dim, d)
        elif fusion_strategy == "add":
            self.fusion_dim = d
            self.fusion_layer = nn.Identity()
        elif fusion_strategy == "gate":
            self.fusion_dim = d
            self.gate_layer = nn.Sequential(nn.Linear(d * 3, d), nn.Sigmoid())
        self.output_projection = nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Dropout(dropout), nn.Linear(d, d))
        self.layer_norms = nn.ModuleDict({k: nn.LayerNorm(d) for k in ("temporal", "spatial", "log", "output")})
        self._keep_ws, self._last_ws = False, None

    def _param_list(self):
        ps = []
        for proj in (self.temporal_projection, self.spatial_projection, self.log_projection):
            ps += [proj.weight, proj.bias]
        for att in (self.temporal_self_attention, self.spatial_self_attention, self.log_self_attention):
            ps += [att.w_v.weight, att.w_v.bias, att.w_o.weight, att.w_o.bias]
        for k in ("temporal", "spatial", "log", "output"):
            ps += [self.layer_norms[k].weight, self.layer_norms[k].bias]
        ps.append(self.modality_weights)
        fa, op = self.feature_attention, self.output_projection
        ps += [fa[0].weight, fa[0].bias, fa[2].weight, fa[2].bias, self.fusion_layer.weight, self.fusion_layer.bias,
               op[0].weight, op[0].bias, op[3].weight, op[3].bias]
        return ps

    def _shape(self, t, s, l) -> _hip.MfaShape:
        sh = _hip.MfaShape()
        sh.B, sh.td, sh.sd, sh.ld = t.shape[0], t.shape[1], s.shape[1], l.shape[1]
        sh.W, sh.heads, sh.training = self.attention_dim, self.num_heads, int(self.training)
        sh.dropout = float(self.output_projection[2].p) if self.training else 0.0
        sh.eps = self.layer_norms["output"].eps
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(t.device).data_ptr() if sh.dropout > 0 else None
        return sh

    def forward(self, temporal_features: torch.Tensor, spatial_features: torch.Tensor, log_features: torch.Tensor,
                return_attention_weights: bool = False):
        if self.fusion_strategy != "concatenate":
            raise NotImplementedError("the HIP path implements fusion_strategy='concatenate' (the one OCEANModel uses)")
        t, s, l = (x.contiguous().float() for x in (temporal_features, spatial_features, log_features))
        if not (t.shape[0] == s.shape[0] == l.shape[0]):
            raise ValueError("temporal, spatial and log features must share the batch dimension")
        if (t.shape[1], s.shape[1], l.shape[1]) != (self.temporal_dim, self.spatial_dim, self.log_dim):
            raise ValueError("feature widths do not match the module's temporal/spatial/log dims")
        self._keep_ws = return_attention_weights
        live = self._param_list()
        dead = []
        for att in (self.temporal_self_attention, self.spatial_self_attention, self.log_self_attention):
            dead += [att.w_q.weight, att.w_q.bias, att.w_k.weight, att.w_k.bias]
        out = _MfaFunction.apply(t, s, l, self, len(live), *live, *dead)
        if not return_attention_weights:
            return out
        B = t.shape[0]
        gates = self._gates(self._last_ws, B)
        self._keep_ws, self._last_ws = False, None
        info = {"modality_weights": torch.softmax(self.modality_weights.detach(), dim=0),
                "feature_importance": {k: gates[i].unsqueeze(1) for i, k in enumerate(("temporal", "spatial", "log"))},
                "self_attention": {k: torch.ones(B, self.num_heads, 1, 1, device=out.device)
                                   for k in ("temporal", "spatial", "log")}}
        return out, info

    def _gates(self, ws, B):
        # the sigmoid feature gates live in the forward workspace right after the hidden layer of feature_attention
        W, al4 = self.attention_dim, lambda n: (n + 3) // 4 * 4
        off = 3 * al4(3 * B * W) + al4(6 * B) + al4(3 * B * W) + al4(6 * B) + al4(3 * B * W) + al4(3 * B * (W // 2))
        return ws[off: off + 3 * B].view(3, B).clone()

    def get_attention_analysis(self, temporal_features, spatial_features, log_features) -> Dict[str, Any]:
        with torch.no_grad():
            _, info = self.forward(temporal_features, spatial_features, log_features, return_attention_weights=True)
        mw = info["modality_weights"].cpu()
        return {"modality_importance": {"temporal": mw[0].item(), "spatial": mw[1].item(), "log": mw[2].item()},
                "feature_importance": {k: v.mean().item() for k, v in info["feature_importance"].items()}}

    def get_model_info(self) -> Dict[str, Any]:
        total = sum(p.numel() for p in self.parameters())
        return {"model_type": "MultiFactorAttention", "temporal_dim": self.temporal_dim, "spatial_dim": self.spatial_dim,
                "log_dim": self.log_dim, "attention_dim": self.attention_dim, "num_heads": self.num_heads,
                "fusion_strategy": self.fusion_strategy, "total_parameters": total,
                "trainable_parameters": sum(p.numel() for p in self.parameters() if p.requires_grad)}
