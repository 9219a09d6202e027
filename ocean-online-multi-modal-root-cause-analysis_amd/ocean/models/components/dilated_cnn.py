"""Temporal encoder: stacked dilated Conv1d blocks over metric windows, on MI355X.

Interface and state_dict keys follow the reference's DilatedCNN /
DilatedConvBlock (ocean/models/components/dilated_cnn.py:16-99, 102-241); the
torch sub-modules held here (nn.Linear / nn.Conv1d / nn.BatchNorm1d) are
parameter containers only - their forward() is never called.  The arithmetic
runs in libocean_hip.so (ocean_dcnn_fwd / ocean_dcnn_bwd): channels-last
activations, every convolution an fp32-MFMA implicit GEMM, BatchNorm
statistics accumulated in the GEMM epilogue.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from ... import _hip

_ACTIVATIONS = ("relu", "gelu", "leaky_relu")


class DilatedConvBlock(nn.Module):
    """conv(k, dilation, 'same' symmetric padding, no bias) -> BatchNorm -> ReLU
    -> Dropout, plus the identity residual (dilated_cnn.py:76-99).  Container
    for one block's parameters; DilatedCNN drives the kernels."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, dilation: int = 1,
                 dropout: float = 0.1, activation: str = "relu"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise ValueError(f"Unsupported activation: {activation}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.dilation = kernel_size, dilation
        self.activation_name = activation
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, dilation=dilation,
                              padding=(kernel_size - 1) * dilation // 2, bias=False)
        self.batch_norm = nn.BatchNorm1d(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.use_residual = in_channels == out_channels
        if not self.use_residual:
            self.residual_conv = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=False)
            self.residual_bn = nn.BatchNorm1d(out_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The block on its own (dilated_cnn.py:76-99): x [batch, in_channels, seq_len] -> [batch, out_channels, seq_len].
        DilatedCNN runs its blocks inside its fused driver instead."""
        if self.activation_name != "relu" or self.kernel_size % 2 == 0:
            raise NotImplementedError("the HIP path implements relu activation and odd kernel sizes")
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected input [batch, {self.in_channels}, seq_len], got {tuple(x.shape)}")
        B, _, T = x.shape
        bns = [self.batch_norm] + ([] if self.use_residual else [self.residual_bn])
        if self.training:
            if B * T < 2:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {torch.Size([B, self.out_channels, T])}")
            if any(bn.momentum is None or not bn.track_running_stats for bn in bns):
                raise NotImplementedError("BatchNorm with running statistics and a fixed momentum")
        params = [self.conv.weight, self.batch_norm.weight, self.batch_norm.bias, self.batch_norm.running_mean,
                  self.batch_norm.running_var]
        if not self.use_residual:
            params += [self.residual_conv.weight, self.residual_bn.weight, self.residual_bn.bias, self.residual_bn.running_mean,
                       self.residual_bn.running_var]
        rows = x.permute(0, 2, 1).contiguous().float().view(B * T, self.in_channels)           # channels last
        out = _ConvBlockFunction.apply(rows, self, B, T, *params)
        if self.training:
            for bn in bns:
                bn.num_batches_tracked += 1
        return out.view(B, T, self.out_channels).permute(0, 2, 1).contiguous()


class _ConvBlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, module, B, T, *params):
        L = _hip.lib()
        sh = _hip.ConvBlockShape()
        sh.B, sh.T, sh.Cin, sh.Cout = B, T, module.in_channels, module.out_channels
        sh.kernel, sh.dilation = module.kernel_size, module.dilation
        sh.training, sh.res_conv = int(module.training), int(not module.use_residual)
        sh.dropout = float(module.dropout.p) if module.training else 0.0
        sh.eps, sh.momentum = float(module.batch_norm.eps), float(module.batch_norm.momentum or 0.0)
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(rows.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        nws = L.ocean_convblock_fwd_workspace(ctypes.byref(sh))
        if nws == 0:
            raise _hip.OceanHipError(f"DilatedConvBlock: unsupported shape (kernel size {module.kernel_size})")
        ws = torch.empty(nws, device=rows.device, dtype=torch.float32)
        out = torch.empty(B * T, sh.Cout, device=rows.device, dtype=torch.float32)
        _hip.check(L.ocean_convblock_fwd(ctypes.byref(sh), _hip.dev_ptr(rows, "x"), _hip.ptr_table(params, "block param"),
                                         _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(out), _hip.stream()), "ocean_convblock_fwd")
        ctx.sh = sh
        ctx.save_for_backward(rows, ws, *params)
        return out

    @staticmethod
    def backward(ctx, dy):
        n_in = 4 + (10 if ctx.sh.res_conv else 5)
        if dy is None:
            return (None,) * n_in
        L = _hip.lib()
        rows, ws, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_convblock_bwd_workspace(ctypes.byref(sh)), device=rows.device, dtype=torch.float32)
        learn = [i for i in range(len(params)) if i % 5 not in (3, 4)]
        got = iter(_hip.zero_grads_like([params[i] for i in learn]))
        grads = [next(got) if i in learn else None for i in range(len(params))]
        dx = torch.empty_like(rows) if ctx.needs_input_grad[0] else None
        _hip.check(L.ocean_convblock_bwd(ctypes.byref(sh), _hip.dev_ptr(rows), _hip.ptr_table(params, "block param"), _hip.dev_ptr(ws),
                                         _hip.dev_ptr(dy.contiguous().float()), _hip.dev_ptr(wb), wb.numel(),
                                         _hip.ptr_table(grads, "block grad"), _hip.dev_ptr(dx), _hip.stream()), "ocean_convblock_bwd")
        return (dx, None, None, None) + tuple(g if (g is not None and ctx.needs_input_grad[4 + i]) else None
                                              for i, g in enumerate(grads))


class _DcnnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        L = _hip.lib()
        shape = module._shape(x)
        B = shape.B
        ws = torch.empty(L.ocean_dcnn_fwd_workspace(ctypes.byref(shape)), device=x.device, dtype=torch.float32)
        out = torch.empty(B, shape.out_dim, device=x.device, dtype=torch.float32)
        ptab = _hip.ptr_table(params, "dcnn param")
        nph = L.ocean_dcnn_num_phases(ctypes.byref(shape))
        sync = module.stats_all_reduce if shape.training else None

        def run(lo, hi):
            _hip.check(L.ocean_dcnn_fwd(ctypes.byref(shape), _hip.dev_ptr(x, "metrics"), ptab, _hip.dev_ptr(ws),
                                        ws.numel(), _hip.dev_ptr(out), lo, hi, _hip.stream()), "ocean_dcnn_fwd")
        if sync is None:
            run(0, nph)
        else:
            for ph in range(nph):
                run(ph, ph + 1)
                if ph < nph - 1:
                    sync(module._stats_view(shape, ws, ph, False))
        ctx.shape, ctx.module = shape, module
        ctx.save_for_backward(x, ws, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _hip.lib()
        x, ws, *params = ctx.saved_tensors
        shape, module = ctx.shape, ctx.module
        dout = dout.contiguous()
        wb = torch.empty(L.ocean_dcnn_bwd_workspace(ctypes.byref(shape)), device=x.device, dtype=torch.float32)
        is_stat = module._stat_slots()
        grads = _hip.zero_grads_like(params, skip=is_stat)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ptab = _hip.ptr_table(params, "dcnn param")
        gtab = _hip.ptr_table(grads, "dcnn grad")
        nph = L.ocean_dcnn_num_phases(ctypes.byref(shape))
        sync = module.stats_all_reduce if shape.training else None

        def run(lo, hi):
            _hip.check(L.ocean_dcnn_bwd(ctypes.byref(shape), _hip.dev_ptr(x), ptab, _hip.dev_ptr(ws), _hip.dev_ptr(wb),
                                        wb.numel(), _hip.dev_ptr(dout, "grad_output"), gtab, _hip.dev_ptr(dx),
                                        lo, hi, _hip.stream()), "ocean_dcnn_bwd")
        start = module.stats_all_reduce_start if shape.training else None
        if sync is None:
            run(0, nph)
        elif start is None:
            for ph in range(nph):
                run(ph, ph + 1)
                if ph < nph - 1:
                    sync(module._stats_view(shape, wb, ph, True))
        else:
            # the statistics block of the layer below is complete after the first part of a phase: its all-reduce
            # travels while the second part (this layer's weight gradient) runs, and is waited for before the next phase
            def part(ph, which):
                _hip.check(L.ocean_dcnn_bwd_part(ctypes.byref(shape), _hip.dev_ptr(x), ptab, _hip.dev_ptr(ws), _hip.dev_ptr(wb),
                                                 wb.numel(), _hip.dev_ptr(dout, "grad_output"), gtab, _hip.dev_ptr(dx),
                                                 ph, which, _hip.stream()), "ocean_dcnn_bwd_part")
            for ph in range(nph):
                part(ph, 0)
                work = start(module._stats_view(shape, wb, ph, True)) if ph < nph - 1 else None
                part(ph, 1)
                if work is not None:
                    work.wait()
        out = [dx, None]
        for i, g in enumerate(grads):
            out.append(g if (g is not None and ctx.needs_input_grad[2 + i]) else None)
        return tuple(out)


class DilatedCNN(nn.Module):
    """DilatedCNN(input_dim, channels, kernel_size, dilation_rates, dropout,
    activation, output_dim, global_pooling): [B,T,F] -> [B,output_dim]
    (dilated_cnn.py:108-116, 195-241)."""

    def __init__(self, input_dim: int, channels: Optional[List[int]] = None, kernel_size: int = 3,
                 dilation_rates: Optional[List[int]] = None, dropout: float = 0.1, activation: str = "relu",
                 output_dim: Optional[int] = None, global_pooling: str = "mean"):
        super().__init__()
        channels = [64, 128, 256, 512] if channels is None else list(channels)
        dilation_rates = [1, 2, 4, 8] if dilation_rates is None else list(dilation_rates)
        if len(channels) != len(dilation_rates):
            raise ValueError("Length of channels and dilation_rates must match")
        if len(channels) > _hip.MAX_CONV_LAYERS:
            raise ValueError(f"at most {_hip.MAX_CONV_LAYERS} dilated blocks are supported")
        self.input_dim, self.channels, self.dilation_rates = input_dim, channels, dilation_rates
        self.kernel_size, self.global_pooling, self.activation_name = kernel_size, global_pooling, activation
        self.dropout_p = float(dropout)

        self.input_projection = nn.Linear(input_dim, channels[0])
        self.conv_layers = nn.ModuleList(
            DilatedConvBlock(c, c, kernel_size, d, dropout, activation) for c, d in zip(channels, dilation_rates))
        self.transition_layers = nn.ModuleList(
            nn.Sequential(nn.Conv1d(channels[i], channels[i + 1], kernel_size=1, bias=False),
                          nn.BatchNorm1d(channels[i + 1]), nn.ReLU(inplace=True), nn.Dropout(dropout))
            for i in range(len(channels) - 1))
        if global_pooling == "attention":
            self.attention_pooling = nn.Sequential(nn.Linear(channels[-1], channels[-1] // 4), nn.ReLU(),
                                                   nn.Linear(channels[-1] // 4, 1), nn.Softmax(dim=-1))
        self.output_dim = output_dim if output_dim is not None else channels[-1]
        self.output_projection = nn.Linear(channels[-1], self.output_dim)

        # data-parallel hook: callable(all-reduce) applied to each fp64 BatchNorm
        # statistics block, and the global number of rows behind those statistics
        self.stats_all_reduce: Optional[Callable[[torch.Tensor], None]] = None
        # optional asynchronous form for the backward pass: returns a handle whose wait() orders the current stream
        # behind the reduction (the weight gradient of the layer above runs in between)
        self.stats_all_reduce_start: Optional[Callable[[torch.Tensor], object]] = None
        self.stats_world_size: int = 1
        # optional callable(local batch) -> samples of this step summed over the ranks, for ranks whose batches differ in
        # size (default: stats_world_size * local batch)
        self.stats_global_rows: Optional[Callable[[int], int]] = None

    # ------------------------------------------------------------------ plumbing
    def _bn_modules(self):
        for blk in self.conv_layers:
            yield blk.batch_norm
        for tr in self.transition_layers:
            yield tr[1]

    def _global_batch(self, local_batch: int) -> int:
        if self.stats_all_reduce is None:
            return local_batch
        if self.stats_global_rows is not None:
            return int(self.stats_global_rows(local_batch))
        return local_batch * self.stats_world_size

    def _param_list(self):
        ps = [self.input_projection.weight, self.input_projection.bias]
        for blk in self.conv_layers:
            bn = blk.batch_norm
            ps += [blk.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        for tr in self.transition_layers:
            ps += [tr[0].weight, tr[1].weight, tr[1].bias, tr[1].running_mean, tr[1].running_var]
        ps += [self.output_projection.weight, self.output_projection.bias]
        return ps

    def _stat_slots(self):
        n = len(self.channels)
        slots = [False, False]
        for _ in range(2 * n - 1):
            slots += [False, False, False, True, True]
        return slots + [False, False]

    def _shape(self, x: torch.Tensor) -> _hip.DcnnShape:
        s = _hip.DcnnShape()
        s.B, s.T, s.F = x.shape
        s.L = len(self.channels)
        for i, (c, d) in enumerate(zip(self.channels, self.dilation_rates)):
            s.channels[i], s.dilations[i] = c, d
        s.kernel_size, s.out_dim = self.kernel_size, self.output_dim
        s.training = int(self.training)
        bn0 = self.conv_layers[0].batch_norm
        s.eps, s.momentum = bn0.eps, (bn0.momentum if bn0.momentum is not None else 0.1)
        s.bn_count = x.shape[1] * self._global_batch(x.shape[0])
        p = self.conv_layers[0].dropout.p if self.training else 0.0
        s.dropout = p
        s.seed = _hip.new_seed(p)
        s.seed_epoch = _hip.seed_epoch(x.device).data_ptr() if p > 0 else None
        return s

    @staticmethod
    def _stats_view(shape, ws, phase, backward):
        off, n = ctypes.c_size_t(0), ctypes.c_int(0)
        _hip.check(_hip.lib().ocean_dcnn_stats_block(ctypes.byref(shape), phase, int(backward),
                                                     ctypes.byref(off), ctypes.byref(n)), "ocean_dcnn_stats_block")
        return ws[off.value: off.value + 2 * n.value].view(torch.float64)

    # ------------------------------------------------------------------ API
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3 or x.shape[2] != self.input_dim:
            raise ValueError(f"expected metrics of shape (batch, seq_len, {self.input_dim}), got {tuple(x.shape)}")
        if self.global_pooling not in ("mean", "max", "attention"):
            raise ValueError(f"Unsupported global pooling: {self.global_pooling}")
        if self.global_pooling != "mean" or self.activation_name != "relu" or self.kernel_size % 2 == 0 \
                or any(not b.use_residual for b in self.conv_layers):
            raise NotImplementedError("the HIP path implements relu activation, mean pooling, odd kernel sizes")
        bsz, seq = x.shape[0], x.shape[1]
        if self.training and seq * self._global_batch(bsz) == 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                             f"{torch.Size([bsz, self.channels[0], seq])}")
        x = x.contiguous().float()
        out = _DcnnFunction.apply(x, self, *self._param_list())
        if self.training:
            with torch.no_grad():
                torch._foreach_add_([bn.num_batches_tracked for bn in self._bn_modules()], 1)   # one launch for all layers
        return out

    def get_temporal_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        """The output of every dilated block pooled over time, one [B, C_i] tensor per block (dilated_cnn.py:243-277;
        the reference pools with max only when global_pooling == 'max', which has no HIP implementation, so: mean).
        Runs the fused forward and pools the block activations it leaves in its workspace; the result carries no
        gradient.  In training mode the BatchNorm running statistics advance, as a forward pass of the reference's
        blocks would advance them."""
        if x.dim() != 3 or x.shape[2] != self.input_dim:
            raise ValueError(f"expected metrics of shape (batch, seq_len, {self.input_dim}), got {tuple(x.shape)}")
        x = x.contiguous().float()
        L = _hip.lib()
        with torch.no_grad():
            shape = self._shape(x)
            ws = torch.empty(L.ocean_dcnn_fwd_workspace(ctypes.byref(shape)), device=x.device, dtype=torch.float32)
            out = torch.empty(shape.B, shape.out_dim, device=x.device, dtype=torch.float32)
            params = self._param_list()
            nph = L.ocean_dcnn_num_phases(ctypes.byref(shape))
            sync = self.stats_all_reduce if shape.training else None
            for lo, hi in ([(0, nph)] if sync is None else [(p, p + 1) for p in range(nph)]):
                _hip.check(L.ocean_dcnn_fwd(ctypes.byref(shape), _hip.dev_ptr(x, "metrics"), _hip.ptr_table(params, "dcnn param"),
                                            _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(out), lo, hi, _hip.stream()), "ocean_dcnn_fwd")
                if sync is not None and hi < nph:
                    sync(self._stats_view(shape, ws, lo, False))
            feats = []
            for i in range(shape.L):
                off, width = ctypes.c_size_t(0), ctypes.c_int(0)
                rc = L.ocean_dcnn_block_output(ctypes.byref(shape), i, ctypes.byref(off), ctypes.byref(width))
                if rc == 2 and i == shape.L - 1:
                    # the last block's activation is pooled by the pass that would have written it: take the means
                    _hip.check(L.ocean_dcnn_pooled_output(ctypes.byref(shape), ctypes.byref(off), ctypes.byref(width)),
                               "ocean_dcnn_pooled_output")
                    feats.append(ws[off.value: off.value + shape.B * width.value].view(shape.B, width.value).clone())
                    continue
                _hip.check(rc, "ocean_dcnn_block_output")
                act = ws[off.value: off.value + shape.B * shape.T * width.value].view(shape.B, shape.T, width.value)
                pooled = torch.empty(shape.B, width.value, device=x.device, dtype=torch.float32)
                _hip.check(L.ocean_mean_mid_f32(_hip.dev_ptr(act), _hip.dev_ptr(pooled), shape.B, shape.T, width.value,
                                                _hip.stream()), "ocean_mean_mid_f32")
                feats.append(pooled)
            if self.training:
                torch._foreach_add_([bn.num_batches_tracked for bn in self._bn_modules()], 1)
        return feats

    def compute_receptive_field(self) -> int:
        return 1 + sum((self.kernel_size - 1) * d for d in self.dilation_rates)

    def get_model_info(self) -> dict:
        total = sum(p.numel() for p in self.parameters())
        return {"model_type": "DilatedCNN", "input_dim": self.input_dim, "channels": self.channels,
                "dilation_rates": self.dilation_rates, "num_layers": len(self.conv_layers), "receptive_field": self.compute_receptive_field(),
                "global_pooling": self.global_pooling, "total_parameters": total,
                "trainable_parameters": sum(p.numel() for p in self.parameters() if p.requires_grad)}
