"""Secondary measurements of the BASELINE configurations other than the headline one (one GPU, synthetic data):

  C1  README example (5 services, batch 1, seq_len 10, logs [1,10,768]): eval forward latency
  P   predict_root_cause on 32 windows of the headline shape: serving latency
  T   Trainer.train_step at batch 32 of the headline shape: samples/s
  C2  online step, 64 services, batch 256, seq_len 64, in-batch InfoNCE: samples/s
  C4  4096 services (CSR GAT, in-degree 8), batch 64 = one GPU's share of the 512 global batch: samples/s
  C5  per-window online updates (64 services, seq_len 64): update latency p50 / p99 for window batches of 8

    python tools/bench_configs.py [C1 P T C2 C4 C5]

Prints one JSON line per configuration and launch mode (eager, replayed), and for C2 / C4 one more replayed line with
the backward products of the large convolutions split into bf16 halves (labelled).  Same synthetic-input recipe as
bench.py.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch  # noqa: E402
import ocean  # noqa: E402
from ocean.data import ServiceGraph  # noqa: E402
from ocean.models.training import OnlineLearner  # noqa: E402

dev = torch.device("cuda", 0)
g = lambda s: torch.Generator().manual_seed(s)


def make(N, S, deg):
    cfg = ocean.default_config()
    cfg.model.num_services = S
    torch.manual_seed(42)
    m = ocean.OCEANModel(cfg).to(dev)
    nf = torch.randn(N, 12, generator=g(7))
    adj = (torch.rand(N, N, generator=g(11)) < deg / N).float()
    adj.fill_diagonal_(0.0)
    return cfg, m, ServiceGraph(adjacency_matrix=adj, node_features=nf)


def inputs(B, T, S):
    metrics = torch.randn(B, T, 12, generator=g(1000)).to(dev)
    logs = torch.randn(B, 768, generator=g(2000)).to(dev)
    labels = torch.zeros(B, S)
    labels[torch.arange(B), torch.randint(0, S, (B,), generator=g(3000))] = 1.0
    return metrics, logs, labels.to(dev)


def timed(fn, warm, n):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return ts


def c1(use_graph=True):
    cfg, m, graph = make(5, 100, 0.0)
    m.eval()
    m.use_hip_graph = use_graph
    metrics, logs = torch.randn(1, 10, 12, generator=g(1)).to(dev), torch.randn(1, 10, 768, generator=g(2)).to(dev)
    with torch.no_grad():
        ts = sorted(timed(lambda: m(metrics, graph, logs), 10, 200))
    return {"config": "C1 README eval forward (5 services, batch 1, seq_len 10)", "metric": "latency", "unit": "ms",
            "p50": ts[len(ts) // 2] * 1e3, "p99": ts[int(len(ts) * 0.99)] * 1e3, "launch": "hip-graph replay" if m._graphed_fwd else "eager"}


def predict(use_graph=True, B=32, N=256, T=128):
    """predict_root_cause on a batch of windows of the headline shape (256 services, 128 timesteps): serving latency"""
    cfg, m, graph = make(N, N, 8.0)
    m.eval()
    m.use_hip_graph = use_graph
    metrics, logs, _ = inputs(B, T, N)
    ts = sorted(timed(lambda: m.predict_root_cause(metrics, graph, logs, top_k=5), 10, 200))
    return {"config": f"predict_root_cause ({N} services, batch {B}, seq_len {T}, top-5)", "metric": "latency", "unit": "ms",
            "p50": ts[len(ts) // 2] * 1e3, "p99": ts[int(len(ts) * 0.99)] * 1e3, "launch": "hip-graph replay" if m._graphed_fwd else "eager"}


def train_cfg(name, N, S, deg, B, T, steps, use_graph=True, split=None):
    prev = ocean.set_split_bf16(split) if split else None
    try:
        r = _train_cfg(name, N, S, deg, B, T, steps, use_graph)
    finally:
        if split:
            ocean.set_split_bf16(prev)
    if split:
        r["arithmetic"] = f"SECONDARY: large convolution products split into bf16 halves ({split})"
    return r


def _train_cfg(name, N, S, deg, B, T, steps, use_graph=True):
    cfg, m, graph = make(N, S, deg)
    m.contrastive_negatives = "in_batch"
    learner = OnlineLearner(m, cfg)
    learner.use_hip_graph = use_graph
    metrics, logs, labels = inputs(B, T, S)
    for _ in range(3):
        learner.process_tensors(metrics, graph, logs, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        learner.process_tensors(metrics, graph, logs, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": name, "metric": "samples/sec fwd+bwd online step", "unit": "samples/s", "value": B / dt, "ms_per_step": dt * 1e3,
            "batch": B, "seq_len": T, "services": N, "launch": "hip-graph replay" if learner._graphed else "eager"}


def trainer_step(use_graph=True, B=32, N=256, T=128, steps=50):
    """Trainer.train_step (LossFunction with consistency and diversity terms) at the reference's default batch size"""
    from datetime import datetime
    from ocean.models.training import Trainer
    cfg, m, graph = make(N, N, 8.0)
    m.contrastive_negatives = "in_batch"
    tr = Trainer(m, cfg, data_loader=None)
    tr.use_hip_graph = use_graph
    metrics, logs, labels = inputs(B, T, N)
    batch = {"metrics": metrics, "logs": logs, "labels": labels, "graphs": [graph] * B, "timestamps": [datetime(2024, 1, 1)] * B}
    for _ in range(4):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": f"Trainer.train_step ({N} services, batch {B}, seq_len {T})", "metric": "samples/sec training step", "unit": "samples/s",
            "value": B / dt, "ms_per_step": dt * 1e3, "launch": "hip-graph replay" if tr._graphed else "eager"}


def c5(use_graph=True):
    cfg, m, graph = make(64, 64, 4.0)
    m.contrastive_negatives = "in_batch"
    learner = OnlineLearner(m, cfg)
    learner.use_hip_graph = use_graph
    metrics, logs, labels = inputs(8, 64, 64)
    ts = sorted(timed(lambda: learner.process_tensors(metrics, graph, logs, labels), 10, 200))
    return {"config": "C5 per-window online update (64 services, seq_len 64, window batch 8, one stream group on one GPU)",
            "metric": "update latency", "unit": "ms", "p50": ts[len(ts) // 2] * 1e3, "p99": ts[int(len(ts) * 0.99)] * 1e3,
            "samples_per_s": 8 / (sum(ts) / len(ts)), "launch": "hip-graph replay" if learner._graphed else "eager"}


which = sys.argv[1:] or ["C1", "P", "T", "C2", "C4", "C5"]
for w in which:
    for use_graph in (False, True):
        if w == "C1":
            r = c1(use_graph)
        elif w == "P":
            r = predict(use_graph)
        elif w == "T":
            r = trainer_step(use_graph)
        elif w == "C2":
            r = train_cfg("C2 online step (64 services, batch 256, seq_len 64, in-batch InfoNCE)", 64, 64, 4.0, 256, 64, 20, use_graph)
        elif w == "C4":
            r = train_cfg("C4 one GPU's share (4096 services, in-degree 8, batch 64 of 512, seq_len 128, in-batch InfoNCE)", 4096, 4096,
                          8.0, 64, 128, 10, use_graph)
        else:
            r = c5(use_graph)
        print(json.dumps(r), flush=True)
    # the two training configurations once more, replayed, with only the backward products split (every parity test
    # passes unchanged under that setting)
    if w == "C2":
        print(json.dumps(train_cfg("C2 online step (64 services, batch 256, seq_len 64, in-batch InfoNCE)", 64, 64, 4.0, 256, 64, 20, True,
                                   split="backward")), flush=True)
    elif w == "C4":
        print(json.dumps(train_cfg("C4 one GPU's share (4096 services, in-degree 8, batch 64 of 512, seq_len 128, in-batch InfoNCE)", 4096,
                                   4096, 8.0, 64, 128, 10, True, split="backward")), flush=True)
