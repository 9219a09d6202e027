"""Debug aid: directional finite differences of the GNN with dropout under a fixed seed, several step sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import synth
from ocean.models.components import DeviceGraph, GraphNeuralNetwork

for D, H, deg, pdrop in [(128, 4, 20.0, 0.3), (128, 4, 20.0, 0.0), (64, 16, 6.0, 0.3), (24, 3, 3.0, 0.3)]:
    N, B = 48, 3
    m = GraphNeuralNetwork(12, D, 2, H, dropout=pdrop, batch_norm=False)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 5))
    m = m.cuda().train()
    g = DeviceGraph.from_adjacency(synth.random_digraph(N, deg, 9), "cuda")
    nf = torch.randn(N, 12, generator=torch.Generator().manual_seed(1)).cuda()
    gw = torch.randn(B, D, generator=torch.Generator().manual_seed(2)).cuda()
    params = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    def f():
        torch.manual_seed(77)
        return (m.forward_graph(nf, g, B).double() * gw.double()).sum()
    for p in params: p.grad = None
    f().backward()
    grads = [p.grad.clone() for p in params]
    print(f"D={D} H={H} deg={deg} p={pdrop}")
    for name, p, gr in zip(names, params, grads):
        gen = torch.Generator().manual_seed(3)
        d = torch.randn(p.shape, generator=gen).cuda() * p.detach().abs().mean().clamp_min(1e-3)
        an = (gr.double() * d.double()).sum().item()
        row = []
        for eps in (3e-2, 1e-2, 3e-3):
            with torch.no_grad():
                p.add_(d, alpha=eps); fp = f().item()
                p.add_(d, alpha=-2 * eps); fm = f().item()
                p.add_(d, alpha=eps)
            row.append((fp - fm) / (2 * eps))
        print(f"  {name:34s} analytic {an:+.5f}  fd {row[0]:+.5f} {row[1]:+.5f} {row[2]:+.5f}")
