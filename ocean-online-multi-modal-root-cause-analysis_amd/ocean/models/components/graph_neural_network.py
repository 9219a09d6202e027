"""Spatial encoder: graph attention over the service-dependency graph, on MI355X.

Interface and state_dict keys follow the reference's GraphNeuralNetwork
(ocean/models/components/graph_neural_network.py:136-302) as OCEANModel builds
it (conv_type='gat', pooling='mean', batch_norm=False, ocean_model.py:77-86).
GATConv here is a parameter container with torch_geometric's parameter names;
the arithmetic (node projection GEMM, edge scores, masked edge softmax,
neighbour aggregation, mean pooling and all of their backward) runs in
libocean_hip.so (ocean_gnn_fwd / ocean_gnn_bwd).

The reference batches B disjoint copies of one graph; here the graph is kept
once as a device CSR (DeviceGraph) and the copies are a replica count, so the
O(B^2) edge-index concatenation of ocean_model.py:163-167 never happens.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional

import torch
import torch.nn as nn

from ... import _hip


class DeviceGraph:
    """CSR pair of one graph on the device, with GATConv's self-loop rule
    applied (existing self loops dropped, one self loop added per node)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, device):
        ei = edge_index.detach().to("cpu", torch.long)
        src, dst = ei[0], ei[1]
        keep = src != dst
        loops = torch.arange(num_nodes, dtype=torch.long)
        src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
        if src.numel() and (int(src.max()) >= num_nodes or int(dst.max()) >= num_nodes or int(src.min()) < 0):
            raise ValueError("edge_index refers to a node outside [0, num_nodes)")
        E = src.numel()
        order_in = torch.argsort(dst, stable=True)           # by target: forward gather
        order_out = torch.argsort(src, stable=True)          # by source: backward gather
        pos_in = torch.empty(E, dtype=torch.long)
        pos_in[order_in] = torch.arange(E)
        zero = torch.zeros(1, dtype=torch.long)
        rp_in = torch.cat([zero, torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)])
        rp_out = torch.cat([zero, torch.cumsum(torch.bincount(src, minlength=num_nodes), 0)])
        i32 = lambda t: t.to(torch.int32).to(device).contiguous()
        self.arrays = [i32(rp_in), i32(src[order_in]), i32(rp_out), i32(dst[order_out]), i32(pos_in[order_out])]
        self.num_nodes, self.num_edges, self.device = num_nodes, E, torch.device(device)
        self.table = (ctypes.c_void_p * 5)(*[a.data_ptr() for a in self.arrays])

    @staticmethod
    def from_adjacency(adj: torch.Tensor, device) -> "DeviceGraph":
        # ServiceGraph.to_edge_index (ocean/data/data_types.py:197-199): row = source, column = target
        adj = adj.detach().to("cpu")
        if adj.is_sparse:                     # edge list straight to CSR: no dense [N, N] matrix for large graphs
            adj = adj.coalesce()
            ei = adj.indices()[:, adj.values() != 0].contiguous()
        else:
            ei = torch.nonzero(adj, as_tuple=False).t().contiguous()
        return DeviceGraph(ei, adj.shape[0], device)

    @staticmethod
    def from_edge_index(edge_index: torch.Tensor, num_nodes: int, device) -> "DeviceGraph":
        """[2, E] (source, target) -> device CSR pair."""
        return DeviceGraph(edge_index, num_nodes, device)

    @staticmethod
    def tensor_key(adj: torch.Tensor):
        """(storage address, version, shape) of an adjacency tensor, dense or sparse COO: what caches key on."""
        base = adj._indices() if adj.is_sparse else adj
        return (base.data_ptr(), adj._version, tuple(adj.shape), bool(adj.is_sparse))


class GATConv(nn.Module):
    """Parameters of torch_geometric.nn.GATConv(in, out, heads, concat, dropout)
    under its names (lin.weight; att_src/att_dst [1,H,C]; bias)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        for w in (self.lin.weight, self.att_src, self.att_dst):       # glorot uniform
            a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
            nn.init.uniform_(w, -a, a)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # torch_geometric 2.3/2.4 checkpoints name the shared projection lin_src (+ an alias lin_dst)
        old = prefix + "lin_src.weight"
        if old in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(old)
            state_dict.pop(prefix + "lin_dst.weight", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        """The layer on its own, as torch_geometric's GATConv(x, edge_index): x [num_nodes, in_channels], edge_index
        [2, E] (source, target) -> [num_nodes, heads * out_channels] (concat) or [num_nodes, out_channels].
        GraphNeuralNetwork runs its layers inside its fused driver instead."""
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"expected x of shape [num_nodes, {self.in_channels}], got {tuple(x.shape)}")
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), x.shape[0], str(x.device))
        cached = getattr(self, "_device_graph", None)
        if cached is None or cached[0] != key:
            cached = (key, DeviceGraph.from_edge_index(edge_index, x.shape[0], x.device))
            self._device_graph = cached
        params = [self.lin.weight, self.att_src, self.att_dst, self.bias]
        return _GatConvFunction.apply(x.contiguous().float(), self, cached[1], *params)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_device_graph", None)                 # device-side cache: rebuilt on the next call
        return state


class _GatConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, graph, *params):
        L = _hip.lib()
        sh = _hip.GatConvShape()
        sh.N, sh.in_dim, sh.H, sh.C, sh.E = x.shape[0], x.shape[1], module.heads, module.out_channels, graph.num_edges
        sh.concat, sh.training = int(module.concat), int(module.training)
        sh.negative_slope = float(module.negative_slope)
        sh.dropout = float(module.dropout) if module.training else 0.0
        sh.seed = _hip.new_seed(sh.dropout)
        sh.seed_epoch = _hip.seed_epoch(x.device).data_ptr() if sh.dropout > 0 else None
        ctx.set_materialize_grads(False)
        nws = L.ocean_gatconv_fwd_workspace(ctypes.byref(sh))
        if nws == 0:
            raise _hip.OceanHipError(f"GATConv: unsupported shape (heads * out_channels = {sh.H * sh.C}, at most 512)")
        ws = torch.empty(nws, device=x.device, dtype=torch.float32)
        out = torch.empty(sh.N, sh.H * sh.C if module.concat else sh.C, device=x.device, dtype=torch.float32)
        _hip.check(L.ocean_gatconv_fwd(ctypes.byref(sh), _hip.dev_ptr(x, "x"), graph.table, _hip.ptr_table(params, "GATConv param"),
                                       _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(out), _hip.stream()), "ocean_gatconv_fwd")
        ctx.sh, ctx.graph = sh, graph
        ctx.save_for_backward(x, ws, *params)
        return out

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 7
        L = _hip.lib()
        x, ws, *params = ctx.saved_tensors
        sh = ctx.sh
        wb = torch.empty(L.ocean_gatconv_bwd_workspace(ctypes.byref(sh)), device=x.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _hip.check(L.ocean_gatconv_bwd(ctypes.byref(sh), _hip.dev_ptr(x), ctx.graph.table, _hip.ptr_table(params, "GATConv param"),
                                       _hip.dev_ptr(ws), _hip.dev_ptr(dy.contiguous().float()), _hip.dev_ptr(wb), wb.numel(),
                                       _hip.ptr_table(grads, "GATConv grad"), _hip.dev_ptr(dx), _hip.stream()), "ocean_gatconv_bwd")
        return (dx, None, None) + tuple(g if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads))


class _GnnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, node_features, module, graph, replicas, pool_n, *params):
        L = _hip.lib()
        s = module._shape(node_features, graph, replicas, pool_n)
        ws = torch.empty(L.ocean_gnn_fwd_workspace(ctypes.byref(s)), device=node_features.device, dtype=torch.float32)
        rows_out = s.R * s.N // (s.pool_n or s.N)
        out = torch.empty(rows_out, s.D, device=node_features.device, dtype=torch.float32)
        ptab = _hip.ptr_table(params, "gnn param")
        _hip.check(L.ocean_gnn_fwd(ctypes.byref(s), _hip.dev_ptr(node_features, "node_features"), graph.table, ptab,
                                   _hip.dev_ptr(ws), ws.numel(), _hip.dev_ptr(out), _hip.stream()), "ocean_gnn_fwd")
        ctx.s, ctx.graph = s, graph
        ctx.save_for_backward(node_features, ws, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _hip.lib()
        nf, ws, *params = ctx.saved_tensors
        s, graph = ctx.s, ctx.graph
        wb = torch.empty(L.ocean_gnn_bwd_workspace(ctypes.byref(s)), device=nf.device, dtype=torch.float32)
        grads = _hip.zero_grads_like(params)
        _hip.check(L.ocean_gnn_bwd(ctypes.byref(s), _hip.dev_ptr(nf), graph.table, _hip.ptr_table(params, "gnn param"),
                                   _hip.dev_ptr(ws), _hip.dev_ptr(wb), wb.numel(), _hip.dev_ptr(dout.contiguous(), "grad_output"),
                                   _hip.ptr_table(grads, "gnn grad"), _hip.stream()), "ocean_gnn_bwd")
        return (None, None, None, None, None) + tuple(g if ctx.needs_input_grad[5 + i] else None for i, g in enumerate(grads))


class GraphNeuralNetwork(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int = 128, num_layers: int = 2, num_heads: int = 4,
                 dropout: float = 0.1, activation: str = "relu", conv_type: str = "gat", pooling: str = "mean",
                 residual: bool = True, batch_norm: bool = True):
        super().__init__()
        if conv_type not in ("gat", "gcn", "sage", "graph"):
            raise ValueError(f"Unsupported conv_type: {conv_type}")
        if activation not in ("relu", "gelu", "leaky_relu"):
            raise ValueError(f"Unsupported activation: {activation}")
        if conv_type != "gat":
            raise NotImplementedError("only conv_type='gat' (the one OCEANModel uses) has a HIP implementation")
        if hidden_dim % num_heads != 0:
            raise ValueError("hidden_dim must be divisible by num_heads")
        if num_layers > 8:
            raise ValueError("at most 8 GAT layers are supported")
        self.input_dim, self.hidden_dim, self.num_layers, self.num_heads = input_dim, hidden_dim, num_layers, num_heads
        self.conv_type, self.pooling, self.residual, self.activation_name = conv_type, pooling, residual, activation
        self.input_projection = nn.Linear(input_dim, hidden_dim)
        self.conv_layers = nn.ModuleList(
            GATConv(hidden_dim, hidden_dim // num_heads, heads=num_heads, dropout=dropout, concat=(i < num_layers - 1))
            for i in range(num_layers))
        self.batch_norms = nn.ModuleList(nn.BatchNorm1d(hidden_dim) for _ in range(num_layers)) if batch_norm else None
        self.dropout = nn.Dropout(dropout)
        if pooling == "attention":
            self.attention_pooling = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 4), nn.ReLU(),
                                                   nn.Linear(hidden_dim // 4, 1))
        self.output_projection = nn.Linear(hidden_dim // num_heads, hidden_dim)
        # True: compute one replica and broadcast whenever dropout is inactive (all replicas are then identical);
        # False: always compute every replica, as the reference does
        self.dedupe_replicas = True

    # ------------------------------------------------------------------ plumbing
    def _param_list(self):
        ps = [self.input_projection.weight, self.input_projection.bias]
        for conv in self.conv_layers:
            ps += [conv.lin.weight, conv.att_src, conv.att_dst, conv.bias]
        return ps + [self.output_projection.weight, self.output_projection.bias]

    def _dropout_p(self) -> float:
        return float(self.dropout.p) if self.training else 0.0

    def _shape(self, node_features, graph: DeviceGraph, replicas: int, pool_n: int) -> _hip.GnnShape:
        s = _hip.GnnShape()
        s.N, s.R, s.G = graph.num_nodes, replicas, node_features.shape[1]
        s.D, s.H, s.layers, s.E, s.pool_n = self.hidden_dim, self.num_heads, self.num_layers, graph.num_edges, pool_n
        s.training = int(self.training)
        s.dropout = self._dropout_p()
        s.negative_slope = self.conv_layers[0].negative_slope
        s.seed = _hip.new_seed(s.dropout)
        s.seed_epoch = _hip.seed_epoch(node_features.device).data_ptr() if s.dropout > 0 else None
        return s

    def _check_supported(self):
        if self.pooling != "mean" or self.batch_norms is not None or self.activation_name != "relu" or not self.residual:
            raise NotImplementedError("the HIP path implements pooling='mean', batch_norm=False, relu, residual=True "
                                      "(OCEANModel's configuration, ocean_model.py:77-86)")

    # ------------------------------------------------------------------ API
    def forward_graph(self, node_features: torch.Tensor, graph: DeviceGraph, batch_size: int) -> torch.Tensor:
        """`batch_size` disjoint copies of `graph` with identical node features
        (the batched graph of ocean_model.py:155-171) -> [batch_size, hidden].
        With dropout inactive every copy is identical: one is computed and
        broadcast; otherwise each copy draws its own masks."""
        self._check_supported()
        nf = node_features.contiguous().float()
        if nf.shape[0] != graph.num_nodes:
            raise ValueError("node_features rows must equal the graph's node count")
        replicas = batch_size if (self._dropout_p() > 0 or not self.dedupe_replicas) else 1
        out = _GnnFunction.apply(nf, self, graph, replicas, 0, *self._param_list())
        return out if replicas == batch_size else out.expand(batch_size, -1)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, batch: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference signature (graph_neural_network.py:236-237): x [n,in],
        edge_index [2,E] over the same n nodes, optional batch vector of
        equal-sized consecutive graphs."""
        self._check_supported()
        n = x.shape[0]
        if batch is None:
            pool_n = 0
        else:
            nb = int(batch.max().item()) + 1
            if n % nb != 0 or not torch.equal(batch.cpu(), torch.arange(nb).repeat_interleave(n // nb)):
                raise NotImplementedError("batch must describe consecutive graphs of equal size")
            pool_n = n // nb
        graph = DeviceGraph(edge_index, n, x.device)
        return _GnnFunction.apply(x.contiguous().float(), self, graph, 1, pool_n, *self._param_list())

    def get_node_embeddings(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        """Node-level output of the last GAT layer, without pooling and output projection
        (graph_neural_network.py:320-356): [n, hidden // heads] as OCEANModel builds the encoder (its last layer
        averages the heads).  Read from the fused forward's workspace; carries no gradient."""
        self._check_supported()
        L = _hip.lib()
        with torch.no_grad():
            x = x.contiguous().float()
            graph = DeviceGraph(edge_index, x.shape[0], x.device)
            s = self._shape(x, graph, 1, 0)
            ws = torch.empty(L.ocean_gnn_fwd_workspace(ctypes.byref(s)), device=x.device, dtype=torch.float32)
            out = torch.empty(1, s.D, device=x.device, dtype=torch.float32)
            _hip.check(L.ocean_gnn_fwd(ctypes.byref(s), _hip.dev_ptr(x, "node_features"), graph.table,
                                       _hip.ptr_table(self._param_list(), "gnn param"), _hip.dev_ptr(ws), ws.numel(),
                                       _hip.dev_ptr(out), _hip.stream()), "ocean_gnn_fwd")
            off, width = ctypes.c_size_t(0), ctypes.c_int(0)
            _hip.check(L.ocean_gnn_layer_output(ctypes.byref(s), s.layers - 1, ctypes.byref(off), ctypes.byref(width)),
                       "ocean_gnn_layer_output")
            return ws[off.value: off.value + x.shape[0] * width.value].view(x.shape[0], width.value).clone()

    def edge_kernel_family(self, graph: DeviceGraph, replicas: int = 1) -> str:
        """Which kernels the edge phase runs on for this graph: 'replica_block' (per-node tables and CSR staged in LDS;
        small service graphs), 'node_per_wave' (CSR read from memory; graphs too large for the LDS tables) or 'generic'."""
        s = _hip.GnnShape()
        s.N, s.R, s.G, s.D, s.H = graph.num_nodes, replicas, self.input_dim, self.hidden_dim, self.num_heads
        s.layers, s.E, s.pool_n = self.num_layers, graph.num_edges, 0
        fam = _hip.lib().ocean_gnn_edge_kernel_family(ctypes.byref(s))
        if fam < 0:
            raise _hip.OceanHipError("bad graph shape")
        return ("generic", "node_per_wave", "replica_block")[fam]

    def get_model_info(self) -> dict:
        total = sum(p.numel() for p in self.parameters())
        return {"model_type": "GraphNeuralNetwork", "input_dim": self.input_dim, "hidden_dim": self.hidden_dim,
                "num_layers": self.num_layers, "num_heads": self.num_heads, "conv_type": self.conv_type,
                "pooling": self.pooling, "residual": self.residual, "total_parameters": total,
                "trainable_parameters": sum(p.numel() for p in self.parameters() if p.requires_grad)}
