"""Autograd wrappers over single C-ABI entry points (losses and small glue).

Each function validates shapes on the host, allocates outputs / workspaces
through torch's caching allocator and enqueues the HIP work on the current
stream.  None of them has a CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _hip


class _InfoNce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, positive, negatives, tau, label_offset, mean_rows):
        L = _hip.lib()
        sh = _hip.InfoNceShape()
        sh.B, sh.P = anchor.shape
        sh.Bp = positive.shape[0]
        sh.K = 0 if negatives is None else negatives.shape[0]
        sh.label_offset, sh.mean_rows, sh.tau = label_offset, mean_rows, tau
        ws = torch.empty(L.ocean_infonce_workspace(ctypes.byref(sh)), device=anchor.device, dtype=torch.float32)
        loss = torch.zeros((), device=anchor.device, dtype=torch.float32)
        _hip.check(L.ocean_infonce_fwd(ctypes.byref(sh), _hip.dev_ptr(anchor, "anchor"), _hip.dev_ptr(positive, "positive"),
                                       _hip.dev_ptr(negatives, "negatives"), _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(loss),
                                       _hip.stream()), "ocean_infonce_fwd")
        ctx.sh = sh
        ctx.save_for_backward(anchor, positive, ws)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        L = _hip.lib()
        anchor, positive, ws = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_infonce_bwd_workspace(ctypes.byref(sh)), device=anchor.device, dtype=torch.float32)
        da, dp = torch.empty_like(anchor), torch.empty_like(positive)
        _hip.check(L.ocean_infonce_bwd(ctypes.byref(sh), _hip.dev_ptr(anchor), _hip.dev_ptr(positive), _hip.dev_ptr(ws),
                                       _hip.dev_ptr(wb), wb.numel(), _hip.dev_ptr(dloss.contiguous().float(), "dloss"),
                                       _hip.dev_ptr(da), _hip.dev_ptr(dp), _hip.stream()), "ocean_infonce_bwd")
        return da, dp, None, None, None, None


class _InfoNceRows(torch.autograd.Function):
    """reduction='none': one loss per anchor row."""

    @staticmethod
    def forward(ctx, anchor, positive, negatives, tau, label_offset):
        L = _hip.lib()
        sh = _hip.InfoNceShape()
        sh.B, sh.P = anchor.shape
        sh.Bp = positive.shape[0]
        sh.K = 0 if negatives is None else negatives.shape[0]
        sh.label_offset, sh.mean_rows, sh.tau = label_offset, 0, tau
        ws = torch.empty(L.ocean_infonce_workspace(ctypes.byref(sh)), device=anchor.device, dtype=torch.float32)
        rows = torch.empty(sh.B, device=anchor.device, dtype=torch.float32)
        _hip.check(L.ocean_infonce_rows_fwd(ctypes.byref(sh), _hip.dev_ptr(anchor, "anchor"), _hip.dev_ptr(positive, "positive"),
                                            _hip.dev_ptr(negatives, "negatives"), _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(rows),
                                            _hip.stream()), "ocean_infonce_rows_fwd")
        ctx.sh = sh
        ctx.save_for_backward(anchor, positive, ws)
        return rows

    @staticmethod
    def backward(ctx, drows):
        L = _hip.lib()
        anchor, positive, ws = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_infonce_bwd_workspace(ctypes.byref(sh)), device=anchor.device, dtype=torch.float32)
        da, dp = torch.empty_like(anchor), torch.empty_like(positive)
        ws = ws.clone()                          # the backward pass rewrites the saved similarities
        _hip.check(L.ocean_infonce_rows_bwd(ctypes.byref(sh), _hip.dev_ptr(anchor), _hip.dev_ptr(positive), _hip.dev_ptr(ws),
                                            _hip.dev_ptr(wb), wb.numel(), _hip.dev_ptr(drows.contiguous().float(), "d_row_loss"),
                                            _hip.dev_ptr(da), _hip.dev_ptr(dp), _hip.stream()), "ocean_infonce_rows_bwd")
        return da, dp, None, None, None


def infonce(anchor: torch.Tensor, positive: torch.Tensor, negatives: Optional[torch.Tensor] = None,
            temperature: float = 0.1, label_offset: int = 0, mean_rows: int = 0, per_row: bool = False) -> torch.Tensor:
    """InfoNCELoss.forward (graph_fusion.py:112-162): the mean over rows, or with `per_row` (reduction='none') one
    value per anchor row.  negatives given: memory-bank branch; None: in-batch branch where row b of
    `anchor` is matched with row label_offset + b of `positive`."""
    if anchor.dim() != 2 or positive.dim() != 2 or anchor.shape[1] != positive.shape[1]:
        raise ValueError("anchor and positive must be [B, dim] with equal dim")
    if negatives is not None:
        if positive.shape[0] != anchor.shape[0]:
            raise ValueError("anchor and positive must have the same batch size")
        negatives = negatives.detach().contiguous().float()
    if per_row:
        return _InfoNceRows.apply(anchor.contiguous().float(), positive.contiguous().float(), negatives, float(temperature),
                                  int(label_offset))
    return _InfoNce.apply(anchor.contiguous().float(), positive.contiguous().float(), negatives, float(temperature),
                          int(label_offset), int(mean_rows))


class _InfoNce3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z, G, tau, label_offset):
        L = _hip.lib()
        sh = _hip.InfoNce3Shape()
        _, sh.B, sh.P = Z.shape
        sh.tau = tau
        sh.Bp = 0 if G is None else G.shape[1]
        sh.label_offset = label_offset
        ws = torch.empty(L.ocean_infonce3_workspace(ctypes.byref(sh)), device=Z.device, dtype=torch.float32)
        if ws.numel() == 0:
            raise ValueError("infonce_pairs: label_offset + local rows must fit the gathered rows")
        losses = torch.zeros(3, device=Z.device, dtype=torch.float32)
        _hip.check(L.ocean_infonce3_fwd(ctypes.byref(sh), _hip.dev_ptr(Z, "projections"), _hip.dev_ptr(G, "gathered projections"),
                                        _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(losses), _hip.stream()), "ocean_infonce3_fwd")
        ctx.sh = sh
        ctx.save_for_backward(ws)
        ctx.shapes = (Z.shape, None if G is None else G.shape)
        return losses

    @staticmethod
    def backward(ctx, dlosses):
        L = _hip.lib()
        (ws,) = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_infonce3_bwd_workspace(ctypes.byref(sh)), device=ws.device, dtype=torch.float32)
        dZ = torch.empty(ctx.shapes[0], device=ws.device, dtype=torch.float32)
        dG = None if ctx.shapes[1] is None else torch.empty(ctx.shapes[1], device=ws.device, dtype=torch.float32)
        _hip.check(L.ocean_infonce3_bwd(ctypes.byref(sh), _hip.dev_ptr(ws), _hip.dev_ptr(wb), wb.numel(),
                                        _hip.dev_ptr(dlosses.contiguous().float(), "dlosses"), _hip.dev_ptr(dZ), _hip.dev_ptr(dG),
                                        _hip.stream()), "ocean_infonce3_bwd")
        return dZ, dG, None, None


def infonce_pairs(projections: torch.Tensor, temperature: float = 0.1, gathered: Optional[torch.Tensor] = None,
                  label_offset: int = 0) -> torch.Tensor:
    """The in-batch InfoNCE losses of the pairs (0,1), (0,2), (1,2) of `projections` [3, B, P] as one [3] tensor: what
    three InfoNCELoss.forward(anchor, positive) calls return (graph_fusion.py:112-162, 231-249), from one normalisation,
    one batched similarity launch and one row pass.  `gathered` [3, B_all, P] (data parallel): the projections of every
    rank, whose rows serve as positives; local row r then pairs with gathered row label_offset + r.  The backward pass
    consumes the saved state: differentiate once."""
    if projections.dim() != 3 or projections.shape[0] != 3:
        raise ValueError("projections must be [3, B, dim]")
    if gathered is not None:
        if gathered.dim() != 3 or gathered.shape[0] != 3 or gathered.shape[2] != projections.shape[2]:
            raise ValueError("gathered must be [3, B_all, dim]")
        gathered = gathered.contiguous().float()
    return _InfoNce3.apply(projections.contiguous().float(), gathered, float(temperature), int(label_offset))


class StackedLosses(dict):
    """A dictionary of scalar losses that may also carry them as ONE tensor, `stacked` [n] in the dictionary's order
    (None when the entries are separate tensors): combine_losses then reads one operand and its backward writes one."""
    stacked: Optional[torch.Tensor] = None


class LossDict(dict):
    """compute_loss's result; `vec` (or None) holds [total, prediction, contrastive] as one device tensor, so that the
    learner's read-back needs no stacking launch."""
    vec: Optional[torch.Tensor] = None


class _CombineLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, stacked, wp, wc):
        out = torch.empty(3, device=pred.device, dtype=torch.float32)
        n = 0 if stacked is None else stacked.numel()
        _hip.check(_hip.lib().ocean_loss_combine_fwd(_hip.dev_ptr(pred, "prediction loss"), _hip.dev_ptr(stacked, "contrastive losses"), n,
                                                     wp, wc, _hip.dev_ptr(out), _hip.stream()), "ocean_loss_combine_fwd")
        ctx.cfg = (wp, wc, n, None if stacked is None else stacked.shape)
        ctx.set_materialize_grads(False)
        total, csum = out[0], out[2]
        ctx.mark_non_differentiable(out)
        return total, csum, out

    @staticmethod
    def backward(ctx, d_total, d_csum, _d_out):
        wp, wc, n, shape = ctx.cfg
        ref = d_total if d_total is not None else d_csum
        if ref is None:
            return None, None, None, None
        d_pred = torch.empty((), device=ref.device, dtype=torch.float32)
        d_c = None if n == 0 else torch.empty(shape, device=ref.device, dtype=torch.float32)
        f = lambda g: None if g is None else g.contiguous().float()
        _hip.check(_hip.lib().ocean_loss_combine_bwd(_hip.dev_ptr(f(d_total)), _hip.dev_ptr(f(d_csum)), n, wp, wc, _hip.dev_ptr(d_pred),
                                                     _hip.dev_ptr(d_c), _hip.stream()), "ocean_loss_combine_bwd")
        return d_pred, d_c, None, None


def combine_losses(pred: torch.Tensor, stacked: Optional[torch.Tensor], wp: float, wc: float):
    """(total, contrastive sum, [total, pred, contrastive sum]) with total = wp * pred + wc * (0 + stacked[0] + ...),
    the arithmetic of OCEANModel.compute_loss (ocean_model.py:257-266) in one launch (one more in backward)."""
    if pred.dim() != 0 or pred.dtype != torch.float32 or (stacked is not None and (stacked.dim() != 1 or stacked.dtype != torch.float32)):
        raise ValueError("combine_losses takes an fp32 scalar and an fp32 vector")
    return _CombineLosses.apply(pred, None if stacked is None else stacked.contiguous(), float(wp), float(wc))


_loss_states = {}


def _loss_state(device) -> torch.Tensor:
    """The zero-initialised ocean_loss_state of `device` (per-block sums and the finished-block counter of
    ocean_pred_total_fwd), made on first use and kept: the kernel leaves the counter at zero."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _loss_states.get(key)
    if t is None:
        t = _loss_states[key] = torch.zeros(512 + 4, device=device, dtype=torch.float32)
    return t


class _PredTotal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, targets, mean_n, stacked, wp, wc):
        out = torch.empty(3, device=probs.device, dtype=torch.float32)
        n_c = 0 if stacked is None else stacked.numel()
        _hip.check(_hip.lib().ocean_pred_total_fwd(_hip.dev_ptr(probs, "probs"), _hip.dev_ptr(targets, "targets"), probs.numel(), mean_n,
                                                   _hip.dev_ptr(stacked, "contrastive losses"), n_c, wp, wc,
                                                   _hip.dev_ptr(_loss_state(probs.device)), _hip.dev_ptr(out), _hip.stream()),
                   "ocean_pred_total_fwd")
        ctx.cfg = (mean_n, wp, wc, n_c, None if stacked is None else stacked.shape)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(probs, targets)
        total, pred, csum = out[0], out[1], out[2]
        ctx.mark_non_differentiable(out)
        return total, pred, csum, out

    @staticmethod
    def backward(ctx, d_total, d_pred, d_csum, _d_out):
        mean_n, wp, wc, n_c, shape = ctx.cfg
        probs, targets = ctx.saved_tensors
        if d_total is None and d_pred is None and d_csum is None:
            return None, None, None, None, None, None
        dp = torch.empty_like(probs)
        d_c = None if n_c == 0 else torch.empty(shape, device=probs.device, dtype=torch.float32)
        f = lambda g: None if g is None else g.contiguous().float()
        _hip.check(_hip.lib().ocean_pred_total_bwd(_hip.dev_ptr(probs), _hip.dev_ptr(targets), _hip.dev_ptr(f(d_total)), _hip.dev_ptr(f(d_csum)),
                                                   _hip.dev_ptr(f(d_pred)), probs.numel(), mean_n, n_c, wp, wc, _hip.dev_ptr(dp),
                                                   _hip.dev_ptr(d_c), _hip.stream()), "ocean_pred_total_bwd")
        return dp, None, None, d_c, None, None


def prediction_and_total(probs: torch.Tensor, targets: torch.Tensor, reduction: str, stacked: Optional[torch.Tensor], wp: float,
                         wc: float):
    """(total, prediction loss, contrastive sum, [total, prediction, contrastive sum]): binary_cross_entropy(probs,
    targets, reduction) followed by combine_losses, as OCEANModel.compute_loss applies them (ocean_model.py:249-266), in
    one launch (one more in backward).  reduction 'mean' or 'sum'."""
    if probs.shape != targets.shape:
        raise ValueError(f"Using a target size ({tuple(targets.shape)}) that is different to the input size "
                         f"({tuple(probs.shape)}) is deprecated. Please ensure they have the same size.")
    if reduction not in ("mean", "sum"):
        raise ValueError(f"{reduction} is not a valid value for reduction")
    if stacked is not None and (stacked.dim() != 1 or stacked.dtype != torch.float32):
        raise ValueError("prediction_and_total takes the contrastive losses as one fp32 vector")
    return _PredTotal.apply(probs.contiguous().float(), targets.to(probs.device).contiguous().float(),
                            0 if reduction == "mean" else 1, None if stacked is None else stacked.contiguous(), float(wp), float(wc))


class _Bce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, targets, mean_n):
        loss = torch.zeros((), device=probs.device, dtype=torch.float32)
        _hip.check(_hip.lib().ocean_bce_fwd(_hip.dev_ptr(probs, "probs"), _hip.dev_ptr(targets, "targets"), probs.numel(),
                                            mean_n, _hip.dev_ptr(loss), _hip.stream()), "ocean_bce_fwd")
        ctx.mean_n = mean_n
        ctx.save_for_backward(probs, targets)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        probs, targets = ctx.saved_tensors
        dp = torch.empty_like(probs)
        _hip.check(_hip.lib().ocean_bce_bwd(_hip.dev_ptr(probs), _hip.dev_ptr(targets), _hip.dev_ptr(dloss.contiguous().float()),
                                            probs.numel(), ctx.mean_n, _hip.dev_ptr(dp), _hip.stream()), "ocean_bce_bwd")
        return dp, None, None


class _BceElem(torch.autograd.Function):
    """reduction='none': the per-element loss, same shape as the input."""

    @staticmethod
    def forward(ctx, probs, targets):
        out = torch.empty_like(probs)
        _hip.check(_hip.lib().ocean_bce_elem_fwd(_hip.dev_ptr(probs, "probs"), _hip.dev_ptr(targets, "targets"), probs.numel(),
                                                 _hip.dev_ptr(out), _hip.stream()), "ocean_bce_elem_fwd")
        ctx.save_for_backward(probs, targets)
        return out

    @staticmethod
    def backward(ctx, dout):
        probs, targets = ctx.saved_tensors
        dp = torch.empty_like(probs)
        _hip.check(_hip.lib().ocean_bce_elem_bwd(_hip.dev_ptr(probs), _hip.dev_ptr(targets), _hip.dev_ptr(dout.contiguous().float()),
                                                 probs.numel(), _hip.dev_ptr(dp), _hip.stream()), "ocean_bce_elem_bwd")
        return dp, None


def binary_cross_entropy(probs: torch.Tensor, targets: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """F.binary_cross_entropy on probabilities (ocean_model.py:249-253), log terms clamped at -100."""
    if probs.shape != targets.shape:
        raise ValueError(f"Using a target size ({tuple(targets.shape)}) that is different to the input size "
                         f"({tuple(probs.shape)}) is deprecated. Please ensure they have the same size.")
    if reduction == "none":
        return _BceElem.apply(probs.contiguous().float(), targets.to(probs.device).contiguous().float())
    if reduction not in ("mean", "sum"):
        raise ValueError(f"{reduction} is not a valid value for reduction")
    loss = _Bce.apply(probs.contiguous().float(), targets.to(probs.device).contiguous().float(),
                      0 if reduction == "mean" else 1)
    return loss


def mean_tokens(x: torch.Tensor) -> torch.Tensor:
    """[B,L,C] -> [B,C] mean over L (log-embedding sequences, ocean_model.py:179-181).  Inputs only: no gradient."""
    if x.dim() != 3:
        raise ValueError("expected [batch, tokens, dim]")
    if x.requires_grad:
        raise NotImplementedError("mean_tokens is applied to model inputs and carries no gradient")
    x = x.contiguous().float()
    out = torch.empty(x.shape[0], x.shape[2], device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().ocean_mean_mid_f32(_hip.dev_ptr(x, "logs"), _hip.dev_ptr(out), x.shape[0], x.shape[1], x.shape[2],
                                             _hip.stream()), "ocean_mean_mid_f32")
    return out


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        loss = torch.zeros((), device=a.device, dtype=torch.float32)
        _hip.check(_hip.lib().ocean_mse_fwd(_hip.dev_ptr(a), _hip.dev_ptr(b), a.numel(), _hip.dev_ptr(loss), _hip.stream()), "ocean_mse_fwd")
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        _hip.check(_hip.lib().ocean_mse_bwd(_hip.dev_ptr(a), _hip.dev_ptr(b), _hip.dev_ptr(dloss.contiguous().float()), a.numel(),
                                            _hip.dev_ptr(da), _hip.stream()), "ocean_mse_bwd")
        return da, None


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss()(a, b) with the gradient flowing to `a` (trainer.py:94: b is the detached previous batch)."""
    if a.shape != b.shape:
        raise ValueError(f"mse_loss needs equal shapes, got {tuple(a.shape)} and {tuple(b.shape)}")
    return _Mse.apply(a.contiguous().float(), b.detach().contiguous().float())


class _Diversity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L = _hip.lib()
        B, E = x.shape
        ws = torch.empty(L.ocean_diversity_workspace(B, E), device=x.device, dtype=torch.float32)
        loss = torch.zeros((), device=x.device, dtype=torch.float32)
        _hip.check(L.ocean_diversity_fwd(_hip.dev_ptr(x), B, E, _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(loss), _hip.stream()),
                   "ocean_diversity_fwd")
        ctx.shape = (B, E)
        ctx.save_for_backward(ws)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (ws,) = ctx.saved_tensors
        B, E = ctx.shape
        dx = torch.empty(B, E, device=ws.device, dtype=torch.float32)
        ws = ws.clone()                      # backward overwrites the similarity matrix; keep double-backward-free reuse safe
        _hip.check(_hip.lib().ocean_diversity_bwd(B, E, _hip.dev_ptr(ws), _hip.dev_ptr(dloss.contiguous().float()), _hip.dev_ptr(dx),
                                                  _hip.stream()), "ocean_diversity_bwd")
        return dx


def diversity_loss(x: torch.Tensor) -> torch.Tensor:
    """mean over i != j of |cos(x_i, x_j)| (trainer.py:97-108)."""
    if x.dim() != 2:
        raise ValueError("expected [batch, dim]")
    return _Diversity.apply(x.contiguous().float())
