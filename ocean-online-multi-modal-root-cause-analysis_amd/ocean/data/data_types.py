"""Argument types of the model / learner API (ocean/data/data_types.py:160-199,
264-282 of the reference): the service graph and one multi-modal sample.
The pandas record types of the reference's ingestion layer are not part of the
model hot path and are not provided."""
from __future__ import annotations

from dataclasses import dataclass
from datetime import datetime
from typing import Any, Dict, List, Optional, Union

import torch


@dataclass
class ServiceGraph:
    adjacency_matrix: torch.Tensor             # [num_services, num_services]; entry [r, c] != 0 is an edge r -> c
    node_features: torch.Tensor                # [num_services, feature_dim]
    edge_weights: Optional[torch.Tensor] = None
    service_names: Optional[List[str]] = None

    def __post_init__(self):
        n_adj, n_feat = self.adjacency_matrix.size(0), self.node_features.size(0)
        if n_adj != n_feat:
            raise ValueError(f"Adjacency matrix size ({n_adj}) does not match node features size ({n_feat})")
        if self.service_names is not None and len(self.service_names) != n_adj:
            raise ValueError(f"Number of service names ({len(self.service_names)}) does not match "
                             f"adjacency matrix size ({n_adj})")

    @property
    def num_services(self) -> int:
        return self.adjacency_matrix.size(0)

    @property
    def feature_dim(self) -> int:
        return self.node_features.size(1)

    def to_edge_index(self) -> torch.Tensor:
        """[2, E] (source row, target row) of the non-zero entries (data_types.py:197-199).  A sparse COO adjacency
        (large service graphs: 4096 services are 64 MB dense, ~0.5 MB as an edge list) gives its stored non-zeros
        in the same row-major order, and no dense matrix is ever formed."""
        adj = self.adjacency_matrix
        if adj.is_sparse:
            adj = adj.coalesce()
            return adj.indices()[:, adj.values() != 0].contiguous()
        return torch.nonzero(adj, as_tuple=False).t().contiguous()

    @staticmethod
    def from_edge_index(edge_index: torch.Tensor, num_services: int, node_features: torch.Tensor,
                        service_names: Optional[List[str]] = None) -> "ServiceGraph":
        """A graph given as an edge list [2, E] (row 0 = source, row 1 = target); the adjacency is kept as a sparse
        COO tensor.  No counterpart in the reference, whose graph builder always emits a dense matrix."""
        ei = edge_index.detach().to("cpu", torch.long)
        adj = torch.sparse_coo_tensor(ei, torch.ones(ei.shape[1]), (num_services, num_services)).coalesce()
        return ServiceGraph(adjacency_matrix=adj, node_features=node_features, service_names=service_names)


@dataclass
class MultiModalFeatures:
    temporal_features: torch.Tensor
    spatial_features: torch.Tensor
    attention_weights: Optional[torch.Tensor] = None
    fused_representation: Optional[torch.Tensor] = None

    def __post_init__(self):
        if self.temporal_features.size(0) != self.spatial_features.size(0):
            raise ValueError("Temporal and spatial features must have the same batch size")


@dataclass
class DatasetSample:
    metrics: torch.Tensor      # [seq_len, num_metrics]
    logs: torch.Tensor         # [log_embedding_dim] or [num_logs, log_embedding_dim]
    graph: ServiceGraph
    label: torch.Tensor        # [num_services]
    timestamp: datetime

    def to_dict(self) -> Dict[str, Any]:
        return {"metrics": self.metrics, "logs": self.logs, "graph": self.graph, "label": self.label,
                "timestamp": self.timestamp}


BatchData = Dict[str, Union[torch.Tensor, List[ServiceGraph], List[datetime]]]
