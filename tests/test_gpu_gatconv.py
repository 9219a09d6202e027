"""GATConv called on its own (torch_geometric's layer as the reference's GNN instantiates it,
graph_neural_network.py:82-90) through the HIP path against the oracle's restatement of the published layer
(oracle.gat_conv; torch_geometric itself is not installed: parity unpinned at this layer).  fp32; outputs 1e-5,
gradients 1e-3."""
import pytest
import torch

from oracle import ocean_ref as R

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("N,E,cin,C,H,concat", [(6, 10, 7, 8, 2, True), (50, 300, 32, 16, 4, True), (50, 300, 32, 16, 4, False),
                                                (256, 2048, 128, 32, 4, True), (1, 0, 5, 4, 1, True), (33, 100, 9, 5, 3, False),
                                                (200, 900, 64, 64, 8, True)])
def test_gatconv_matches_the_oracle(N, E, cin, C, H, concat):
    from ocean.models.components.graph_neural_network import GATConv
    torch.manual_seed(N + E)
    m = GATConv(cin, C, heads=H, concat=concat, dropout=0.0)
    with torch.no_grad():
        m.bias.uniform_(-0.5, 0.5)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda().train()
    x = torch.randn(N, cin)
    ei = torch.randint(0, N, (2, E))                       # duplicates and self loops included
    xr = x.clone().requires_grad_(True)
    yr = R.gat_conv(xr, ei, sd["lin.weight"], sd["att_src"], sd["att_dst"], sd["bias"], H, concat)
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = m(xg, ei.cuda())
    close(y, yr, 1e-5, "output")
    (y * w.cuda()).sum().backward()
    close(xg.grad, xr.grad, 1e-3, "dx")
    for name, p in m.named_parameters():
        close(p.grad, sd[name].grad, 1e-3, name)
    # another edge list on the same module: the cached device graph is replaced
    ei2 = torch.randint(0, N, (2, max(E // 2, 1)))
    close(m(x.cuda(), ei2.cuda()), R.gat_conv(x, ei2, sd["lin.weight"], sd["att_src"], sd["att_dst"], sd["bias"], H, concat), 1e-5,
          "second graph")


def test_gatconv_attention_dropout_and_errors():
    from ocean.models.components.graph_neural_network import GATConv
    torch.manual_seed(1)
    m = GATConv(16, 8, heads=4, dropout=0.5).cuda().train()
    x, ei = torch.randn(40, 16).cuda().requires_grad_(True), torch.randint(0, 40, (2, 200)).cuda()
    y1, y2 = m(x, ei), m(x, ei)
    assert not torch.equal(y1, y2) and bool(torch.isfinite(y1).all())
    y1.square().mean().backward()
    assert bool(torch.isfinite(x.grad).all()) and all(p.grad is not None for p in m.parameters())
    m.eval()
    assert torch.equal(m(x, ei), m(x, ei))
    with pytest.raises(RuntimeError):
        m(torch.randn(40, 15).cuda(), ei)
    with pytest.raises(ValueError):
        m(x, torch.tensor([[0, 41], [1, 2]]).cuda())
