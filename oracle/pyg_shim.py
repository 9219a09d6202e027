"""Import shims used ONLY by oracle/gen_golden.py in the build container.

The reference imports three packages that are not installed (torch_geometric,
wandb, seaborn).  wandb / seaborn are never called on the model hot path, so
empty modules satisfy the import.  torch_geometric's GATConv IS on the path:
the shim class below holds GATConv's parameters under PyG's names and calls
oracle.ocean_ref.gat_conv - i.e. every fixture that flows through the GNN is
pinned against the reference's own GraphNeuralNetwork/OCEANModel code *around*
GATConv, while the GATConv arithmetic itself stays "parity unpinned".
"""
from __future__ import annotations

import importlib.machinery
import math
import sys
import types

import torch
import torch.nn as nn

from . import ocean_ref


class GATConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2,
                 dropout=0.0, add_self_loops=True, bias=True, **kw):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        for w in (self.lin.weight, self.att_src, self.att_dst):   # glorot
            fan = w.size(-2) + w.size(-1)
            a = math.sqrt(6.0 / fan)
            nn.init.uniform_(w, -a, a)

    def forward(self, x, edge_index):
        assert not (self.training and self.dropout > 0), "shim has no attention dropout"
        return ocean_ref.gat_conv(x, edge_index, self.lin.weight, self.att_src, self.att_dst,
                                  self.bias, self.heads, self.concat, self.negative_slope)


def _pool(reduce):
    def f(x, batch):
        nb = int(batch.max()) + 1
        if reduce == "mean":
            s = torch.zeros(nb, x.shape[1], dtype=x.dtype).index_add(0, batch, x)
            c = torch.zeros(nb, dtype=x.dtype).index_add(0, batch, torch.ones_like(batch, dtype=x.dtype))
            return s / c[:, None]
        if reduce == "add":
            return torch.zeros(nb, x.shape[1], dtype=x.dtype).index_add(0, batch, x)
        out = torch.full((nb, x.shape[1]), -float("inf"), dtype=x.dtype)
        return out.scatter_reduce(0, batch[:, None].expand_as(x), x, "amax")
    return f


class _Unavailable(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the OCEAN model path")


def install() -> None:
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    for name in ("wandb", "seaborn"):
        if name not in sys.modules:
            mod(name)
    tg = mod("torch_geometric")
    tg.__path__ = []
    tg.nn = mod("torch_geometric.nn", GATConv=GATConv, GraphConv=_Unavailable, GCNConv=_Unavailable,
                SAGEConv=_Unavailable, global_mean_pool=_pool("mean"), global_max_pool=_pool("max"),
                global_add_pool=_pool("add"))
    tg.data = mod("torch_geometric.data", Data=type("Data", (), {}), Batch=type("Batch", (), {}))
