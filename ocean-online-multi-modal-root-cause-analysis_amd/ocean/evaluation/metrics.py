"""Ranking metrics over accumulated root-cause scores, kept on the device.

Mirrors the part of the reference's PerformanceMetrics that consumes the model's
`root_cause_probs` (ocean/evaluation/metrics.py:25-98 accumulation, :119-185
compute_ranking_metrics): every (sample, service) score of every batch goes
into ONE global ranking; precision / recall / hit rate / NDCG at k, and the mean
of precision-at-hit over all relevant entries.

The reference flattens each batch into Python lists on the host (B x S list
appends per batch) and ranks with numpy at the end.  Here a batch is kept as the
device tensors it already is, and the ranking is one device sort plus prefix
sums in float64; only the handful of result scalars cross to the host.  Same
dictionary keys and values.  Entries with exactly equal scores are ranked in
arrival order (a stable sort); numpy's default argsort leaves their order
unspecified, so results can differ from the reference only inside a tie.

The classification / plotting / statistical-test parts of the reference class
(sklearn, scipy, matplotlib) are not part of this package.
"""
from __future__ import annotations

from datetime import datetime
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

ArrayLike = Union[torch.Tensor, np.ndarray]


class PerformanceMetrics:
    def __init__(self, device: Optional[Union[str, torch.device]] = None):
        """device: where the accumulated scores live; default: wherever the first batch is."""
        self.device = torch.device(device) if device is not None else None
        self.reset()

    def reset(self) -> None:
        self._pred: List[torch.Tensor] = []
        self._truth: List[torch.Tensor] = []
        self._score: List[torch.Tensor] = []
        self.timestamps: List[datetime] = []
        self.service_metadata: List[Dict[str, Any]] = []

    def _flat(self, a: ArrayLike) -> torch.Tensor:
        t = torch.as_tensor(a)
        if self.device is None:
            self.device = t.device
        return t.detach().to(self.device, non_blocking=True).reshape(-1)

    def add_predictions(self, predictions: ArrayLike, ground_truth: ArrayLike, prediction_scores: Optional[ArrayLike] = None,
                        service_metadata: Optional[Dict[str, Any]] = None) -> None:
        """One batch: binary predictions, 0/1 labels and (optionally) the scores they were thresholded from; without
        scores the predictions themselves are ranked, as in the reference (metrics.py:66-70)."""
        p = self._flat(predictions)
        self._pred.append(p)
        self._truth.append(self._flat(ground_truth))
        self._score.append(self._flat(prediction_scores) if prediction_scores is not None else p)
        self.timestamps.append(datetime.now())           # one stamp per batch (the reference repeats it per element)
        if service_metadata:
            self.service_metadata.append(service_metadata)

    def __len__(self) -> int:
        return int(sum(t.numel() for t in self._pred))

    # the reference exposes the flattened history as lists; these views serve code that reads them
    @property
    def predictions(self) -> np.ndarray:
        return torch.cat(self._pred).cpu().numpy() if self._pred else np.zeros(0)

    @property
    def ground_truth(self) -> np.ndarray:
        return torch.cat(self._truth).cpu().numpy() if self._truth else np.zeros(0)

    @property
    def prediction_scores(self) -> np.ndarray:
        return torch.cat(self._score).cpu().numpy() if self._score else np.zeros(0)

    def compute_ranking_metrics(self, k_values: List[int] = [1, 3, 5, 10]) -> Dict[str, float]:
        if len(self) == 0:
            return {}
        truth = torch.cat(self._truth).to(torch.float64)
        score = torch.cat(self._score).to(torch.float64)
        n = truth.numel()
        order = torch.sort(score, descending=True, stable=True).indices
        st = truth[order]                                   # labels in ranked order
        hits = torch.cumsum(st, 0)                          # relevant entries among the top i+1
        total = hits[-1]
        ks = [k for k in k_values if k <= n]
        kmax = max(ks) if ks else 0
        disc = 1.0 / torch.log2(torch.arange(2, kmax + 2, device=truth.device, dtype=torch.float64)) if kmax else None
        ideal = torch.sort(truth, descending=True).values[:kmax] if kmax else None
        parts, names = [total.reshape(1)], ["_total"]
        for k in ks:
            parts += [hits[k - 1].reshape(1), (st[:k] * disc[:k]).sum().reshape(1), (ideal[:k] * disc[:k]).sum().reshape(1)]
            names += [f"hits@{k}", f"dcg@{k}", f"idcg@{k}"]
        # mean over the relevant entries of precision at their rank (metrics.py:158-167); entries are 0/1 labels
        rank = torch.arange(1, n + 1, device=truth.device, dtype=torch.float64)
        rel = st == 1
        parts.append(((hits / rank) * rel).sum().reshape(1))
        parts.append(rel.sum().to(torch.float64).reshape(1))
        names += ["_ap_sum", "_ap_n"]
        vals = dict(zip(names, torch.cat(parts).cpu().tolist()))          # the one device -> host read
        out: Dict[str, float] = {}
        for k in ks:
            out[f"precision@{k}"] = vals[f"hits@{k}"] / k
            if vals["_total"] > 0:
                out[f"recall@{k}"] = vals[f"hits@{k}"] / vals["_total"]
            out[f"hit_rate@{k}"] = 1.0 if vals[f"hits@{k}"] > 0 else 0.0
        if vals["_total"] > 0 and vals["_ap_n"] > 0:
            out["mean_average_precision"] = vals["_ap_sum"] / vals["_ap_n"]
        for k in ks:
            if vals[f"idcg@{k}"] > 0:
                out[f"ndcg@{k}"] = vals[f"dcg@{k}"] / vals[f"idcg@{k}"]
        return out
