"""Deterministic synthetic parameters and inputs for fixtures and parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/ocean_ref.py).  The golden generator and
the tests both call these so that large parameter sets never need to be
stored in fixtures: only their outputs are.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def fill_state_dict(sd: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """Overwrite every entry of `sd` (in key order) with seeded values that
    exercise every term: non-unit norm scales, non-zero biases, non-trivial
    BatchNorm running statistics."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or k.endswith("memory_ptr"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            out[k] = 1.0 + 0.2 * torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("memory_bank"):
            out[k] = torch.randn(v.shape, generator=g)
        elif k.endswith("modality_weights"):
            out[k] = 1.0 + 0.3 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) / math.sqrt(max(fan_in, 1))
        elif k.endswith("weight"):            # 1-D weight: a norm scale
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:                                   # bias
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        out[k] = out[k].to(v.dtype)
    return out


def random_digraph(n: int, avg_degree: float, seed: int, self_loop_at: int = -1,
                   isolated: int = -1) -> torch.Tensor:
    """Directed Erdos-Renyi adjacency [n,n] fp32 with zero diagonal; optional
    pre-existing self loop and an isolated node (edge cases GATConv's
    self-loop handling must survive)."""
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand(n, n, generator=g) < avg_degree / max(n, 1)).float()
    adj.fill_diagonal_(0.0)
    if isolated >= 0:
        adj[isolated, :] = 0.0
        adj[:, isolated] = 0.0
    if self_loop_at >= 0:
        adj[self_loop_at, self_loop_at] = 1.0
    return adj


def digest(t: torch.Tensor, n: int = 16) -> torch.Tensor:
    """Small fingerprint of a tensor: [sum, l2 norm, abs-max, first n, n strided]."""
    f = t.detach().double().flatten()
    head = f[:n]
    stride = max(f.numel() // n, 1)
    samp = f[::stride][:n]
    pad = lambda x: torch.cat([x, torch.zeros(n - x.numel(), dtype=torch.double)])
    return torch.cat([torch.stack([f.sum(), f.norm(), f.abs().max()]), pad(head), pad(samp)]).float()


def sample_positions(numel: int, n: int = 256) -> torch.Tensor:
    """Up to n element positions spread over a tensor of `numel` elements: a fixed stride plus a half-stride offset,
    so that a tensor's samples cover its whole range and differ from the positions `digest` looks at."""
    if numel <= n:
        return torch.arange(numel)
    stride = numel // n
    return (torch.arange(n) * stride + stride // 2).clamp_(max=numel - 1)


def samples(t: torch.Tensor, n: int = 256) -> torch.Tensor:
    """The values of `t` at sample_positions(t.numel(), n): what the default-architecture fixtures keep of every
    gradient and post-step tensor for an ELEMENTWISE comparison (whole tensors of the 3.3 M-parameter model would
    make the fixtures tens of megabytes)."""
    f = t.detach().flatten()
    return f[sample_positions(f.numel(), n)].clone()
