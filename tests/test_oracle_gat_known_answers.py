"""Hand-derived known answers for the oracle's GATConv restatement (the one
function without a reference fixture: torch_geometric is a third-party
dependency that is not installed, requirements.txt:3)."""
import torch

from oracle import ocean_ref as R


def params(n_in, H, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(H * C, n_in, generator=g), torch.randn(1, H, C, generator=g),
            torch.randn(1, H, C, generator=g), torch.randn(H * C, generator=g))


def test_identity_adjacency_only_self_loops_alpha_is_one():
    # every node has exactly one incoming edge (its self loop) => alpha = 1 => out = W x + b
    w, a_s, a_d, b = params(5, 2, 3)
    x = torch.randn(4, 5)
    ei = R.edge_index_of(torch.eye(4))
    out = R.gat_conv(x, ei, w, a_s, a_d, b, heads=2, concat=True)
    assert torch.allclose(out, x @ w.t() + b, atol=1e-6)
    out_mean = R.gat_conv(x, ei, w, a_s, a_d, b[:3], heads=2, concat=False)
    assert torch.allclose(out_mean, (x @ w.t()).view(4, 2, 3).mean(1) + b[:3], atol=1e-6)


def test_zero_attention_vectors_give_plain_mean_over_in_neighbours_and_self():
    w, _, _, b = params(3, 1, 4)
    z = torch.zeros(1, 1, 4)
    x = torch.randn(3, 3)
    adj = torch.tensor([[0., 1., 1.], [0., 0., 1.], [0., 0., 0.]])      # 0->1, 0->2, 1->2
    out = R.gat_conv(x, R.edge_index_of(adj), w, z, z, b, heads=1, concat=True)
    wx = x @ w.t()
    expect = torch.stack([wx[0], (wx[0] + wx[1]) / 2, (wx[0] + wx[1] + wx[2]) / 3]) + b
    assert torch.allclose(out, expect, atol=1e-6)


def test_existing_self_loop_is_not_double_counted():
    w, a_s, a_d, b = params(3, 2, 2)
    x = torch.randn(3, 3)
    adj = torch.tensor([[0., 1., 0.], [0., 0., 0.], [0., 0., 0.]])
    adj_loop = adj.clone()
    adj_loop[1, 1] = 1.0
    o1 = R.gat_conv(x, R.edge_index_of(adj), w, a_s, a_d, b, 2, True)
    o2 = R.gat_conv(x, R.edge_index_of(adj_loop), w, a_s, a_d, b, 2, True)
    assert torch.allclose(o1, o2, atol=1e-7)


def test_two_node_softmax_by_hand():
    # target 1 receives from 0 and from itself; alpha = softmax(leaky_relu(s_j + d_1))
    w = torch.eye(2)
    a_s = torch.tensor([[[1.0, 0.0]]])
    a_d = torch.tensor([[[0.0, 1.0]]])
    b = torch.zeros(2)
    x = torch.tensor([[2.0, -1.0], [-3.0, 0.5]])
    adj = torch.tensor([[0., 1.], [0., 0.]])
    out = R.gat_conv(x, R.edge_index_of(adj), w, a_s, a_d, b, 1, True)
    lrelu = lambda v: v if v > 0 else 0.2 * v
    e01, e11 = lrelu(2.0 + 0.5), lrelu(-3.0 + 0.5)
    a01 = torch.exp(torch.tensor(e01)) / (torch.exp(torch.tensor(e01)) + torch.exp(torch.tensor(e11)))
    expect1 = a01 * x[0] + (1 - a01) * x[1]
    assert torch.allclose(out[1], expect1, atol=1e-6)
    assert torch.allclose(out[0], x[0], atol=1e-6)          # node 0: self loop only
