"""Development aid: eager vs HIP-graph update steps on a small configuration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch, ocean
from ocean.data import ServiceGraph
from ocean.models.training import OnlineLearner
import logging
logging.basicConfig(level=logging.WARNING)
g = lambda s: torch.Generator().manual_seed(s)
dev = torch.device("cuda", 0)
N = S = 64; B = int(sys.argv[1]) if len(sys.argv) > 1 else 8; T = 64
def make(use_graph, drop0):
    cfg = ocean.default_config(); cfg.model.num_services = S
    torch.manual_seed(42)
    m = ocean.OCEANModel(cfg).to(dev); m.contrastive_negatives = "in_batch"
    if drop0:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    l = OnlineLearner(m, cfg); l.use_hip_graph = use_graph
    return m, l
nf = torch.randn(N, 12, generator=g(7)); adj = (torch.rand(N, N, generator=g(11)) < 4.0 / N).float(); adj.fill_diagonal_(0)
graph = ServiceGraph(adjacency_matrix=adj, node_features=nf)
def batch(i):
    me = torch.randn(B, T, 12, generator=g(100 + i)).to(dev); lo = torch.randn(B, 768, generator=g(200 + i)).to(dev)
    la = torch.zeros(B, S); la[torch.arange(B), torch.randint(0, S, (B,), generator=g(300 + i))] = 1.0
    return me, lo, la.to(dev)
res = {}
for mode in (False, True):
    m, l = make(mode, True)
    out = [l.process_tensors(*(lambda b: (b[0], graph, b[1], b[2]))(batch(i)))["batch_loss"] for i in range(8)]
    res[mode] = (out, {k: v.detach().clone() for k, v in m.state_dict().items()})
    print("graph" if mode else "eager", ["%.6f" % x for x in out], "graphs recorded:", len(l._graphed))
worst = max((res[False][1][k].float() - res[True][1][k].float()).abs().max().item() / max(res[False][1][k].float().abs().max().item(), 1e-3) for k in res[False][1])
print("max relative state difference after 8 steps (dropout off):", worst)
for mode in (False, True):
    m, l = make(mode, False)
    b = batch(0)
    for _ in range(5): l.process_tensors(b[0], graph, b[1], b[2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = [l.process_tensors(b[0], graph, b[1], b[2])["batch_loss"] for _ in range(50)]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("graph" if mode else "eager", "dropout on: %.3f ms/step" % (dt * 1e3), "losses", ["%.5f" % x for x in losses[:4]])
