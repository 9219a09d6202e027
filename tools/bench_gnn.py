"""GNN (GAT) forward + backward in isolation at the headline shape (256 services, in-degree 8, 1024 replicas, dropout
0.1): device-event time per pass.  usage: python tools/bench_gnn.py [N] [B] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean.models.components import DeviceGraph, GraphNeuralNetwork

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = lambda s: torch.Generator().manual_seed(s)
torch.manual_seed(42)
m = GraphNeuralNetwork(12, 128, 2, 4, dropout=0.1, batch_norm=False).cuda().train()
adj = (torch.rand(N, N, generator=g(11)) < 8.0 / N).float()
adj.fill_diagonal_(0.0)
dg = DeviceGraph.from_adjacency(adj, "cuda")
nf = torch.randn(N, 12, generator=g(7)).cuda()
gw = torch.randn(B, 128, generator=g(2)).cuda()


def step():
    for p in m.parameters():
        p.grad = None
    (m.forward_graph(nf, dg, B) * gw).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
print(f"GNN fwd+bwd N={N} B={B}: {e0.elapsed_time(e1) / iters:.3f} ms")
