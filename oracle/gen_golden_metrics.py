"""Build-container only: records what the reference's PerformanceMetrics.compute_ranking_metrics
(ocean/evaluation/metrics.py:119-185) returns for seeded inputs into tests/golden/ranking_metrics.json.
The reference module is imported by file path with an empty `seaborn` standing in for the missing package
(imported at module level there, never used by the ranking code).  Test infrastructure, not product code.

    python oracle/gen_golden_metrics.py
"""
import importlib.machinery
import importlib.util
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OCEAN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def cases():
    """(name, batches of (scores [B,S], labels [B,S]), k_values); scores are distinct floats: no ties."""
    out = []
    for seed, B, S, nb, pos in ((0, 4, 6, 1, 1), (1, 8, 16, 3, 1), (2, 5, 7, 2, 2), (3, 1, 3, 1, 1), (4, 6, 5, 2, 0)):
        rng = np.random.default_rng(seed)
        batches = []
        for _ in range(nb):
            scores = rng.permutation(B * S).reshape(B, S).astype(np.float64) / (B * S) + rng.uniform(0, 1e-3 / (B * S))
            labels = np.zeros((B, S))
            for b in range(B):
                labels[b, rng.choice(S, size=pos, replace=False)] = 1.0
            batches.append((scores, labels))
        out.append((f"seed{seed}_B{B}_S{S}_x{nb}_pos{pos}", batches, [1, 3, 5, 10]))
    return out


def main():
    if "seaborn" not in sys.modules:
        stub = types.ModuleType("seaborn")
        stub.__spec__ = importlib.machinery.ModuleSpec("seaborn", None)
        sys.modules["seaborn"] = stub
    spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(REF, "ocean/evaluation/metrics.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    golden = {}
    for name, batches, ks in cases():
        pm = ref.PerformanceMetrics()
        for scores, labels in batches:
            pm.add_predictions((scores > 0.5).astype(np.float64), labels, scores)
        golden[name] = {"k_values": ks, "batches": [{"scores": s.tolist(), "labels": l.tolist()} for s, l in batches],
                        "expected": {k: float(v) for k, v in pm.compute_ranking_metrics(ks).items()},
                        "expected_key_order": list(pm.compute_ranking_metrics(ks).keys())}
    with open(os.path.join(REPO, "tests", "golden", "ranking_metrics.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print("wrote", len(golden), "cases")


if __name__ == "__main__":
    main()
