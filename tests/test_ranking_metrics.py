"""Ranking metrics over accumulated root-cause scores: the oracle restatement and the device-side accumulator against
values recorded from the reference's PerformanceMetrics (ocean/evaluation/metrics.py:119-185; fixture written by
oracle/gen_golden_metrics.py; the reference's own tests only assert that the keys exist, tests/unit/test_evaluation.py:64-68)."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ranking_metrics.json")


def _cases():
    with open(GOLDEN) as f:
        return json.load(f)


def _check(got, case):
    exp = case["expected"]
    assert list(got.keys()) == case["expected_key_order"]
    for k, v in exp.items():
        assert abs(got[k] - v) <= 1e-12 * max(1.0, abs(v)), (k, got[k], v)


def test_oracle_ranking_metrics_match_the_reference_values():
    from oracle import ocean_ref as R
    for name, case in _cases().items():
        scores = np.concatenate([np.asarray(b["scores"]).reshape(-1) for b in case["batches"]])
        labels = np.concatenate([np.asarray(b["labels"]).reshape(-1) for b in case["batches"]])
        _check(R.ranking_metrics(labels, scores, case["k_values"]), case)
    assert R.ranking_metrics([], [], [1]) == {}


def _run(device):
    from ocean.evaluation import PerformanceMetrics
    for name, case in _cases().items():
        pm = PerformanceMetrics()
        for b in case["batches"]:
            s = torch.tensor(b["scores"], dtype=torch.float64, device=device)
            y = torch.tensor(b["labels"], dtype=torch.float32, device=device)
            pm.add_predictions((s > 0.5).float(), y, s)
        _check(pm.compute_ranking_metrics(case["k_values"]), case)
        assert len(pm) == sum(len(b["scores"]) * len(b["scores"][0]) for b in case["batches"])
        assert pm.prediction_scores.shape == (len(pm),) and pm.ground_truth.sum() == sum(np.sum(b["labels"]) for b in case["batches"])
    pm = PerformanceMetrics()
    assert pm.compute_ranking_metrics() == {}
    # without scores the predictions are ranked (metrics.py:66-70); numpy inputs are accepted
    pm.add_predictions(np.array([[1.0, 0.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    r = pm.compute_ranking_metrics([1, 3, 5])
    assert r["precision@1"] == 1.0 and r["hit_rate@3"] == 1.0 and "precision@5" not in r and r["ndcg@3"] == 1.0
    pm.reset()
    assert len(pm) == 0


def test_accumulator_on_host_tensors_matches_the_reference_values():
    _run("cpu")


@pytest.mark.gpu
def test_accumulator_on_the_device_matches_the_reference_values_and_a_large_random_case():
    from ocean.evaluation import PerformanceMetrics
    from oracle import ocean_ref as R
    _run("cuda")
    # C3-sized output: 4 batches of [1024, 256] float32 probabilities (distinct with overwhelming probability)
    g = torch.Generator().manual_seed(9)
    pm, all_s, all_y = PerformanceMetrics(), [], []
    for _ in range(4):
        s = torch.rand(1024, 256, generator=g)
        y = torch.zeros(1024, 256)
        y[torch.arange(1024), torch.randint(0, 256, (1024,), generator=g)] = 1.0
        pm.add_predictions((s > 0.5).float().cuda(), y.cuda(), s.cuda())
        all_s.append(s.double().numpy().reshape(-1)); all_y.append(y.numpy().reshape(-1))
    got = pm.compute_ranking_metrics([1, 3, 5, 10])
    ref = R.ranking_metrics(np.concatenate(all_y), np.concatenate(all_s), [1, 3, 5, 10])
    assert list(got.keys()) == list(ref.keys())
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
