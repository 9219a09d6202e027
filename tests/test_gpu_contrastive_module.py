"""ProjectionHead and ContrastiveLearningModule called on their own (the reference exports both,
ocean/models/components/__init__.py, and ContrastiveLearningModule.forward is graph_fusion.py:204-245), through the HIP
path against the oracle.  fp32; outputs 1e-5, losses and gradients 1e-3."""
import itertools

import pytest
import torch
import torch.nn.functional as F

from oracle import ocean_ref as R

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("B,din,hidden,dout", [(1, 24, None, 16), (5, 40, None, 32), (64, 256, None, 128), (300, 100, 72, 20),
                                               (1024, 256, None, 128)])
@pytest.mark.parametrize("training", [False, True])
def test_projection_head_matches_oracle(B, din, hidden, dout, training):
    from ocean.models.components import ProjectionHead
    torch.manual_seed(B + din)
    head = ProjectionHead(din, dout, hidden_dim=hidden, dropout=0.0 if training else 0.1)   # eval: dropout is the identity
    head.train(training)
    sd = {"h." + k: v.detach().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    head = head.cuda()
    x = torch.randn(B, din)
    w = torch.randn(B, dout)
    xr = x.clone().requires_grad_(True)
    yr = R.projection_head(sd, "h.", xr) if sd.get("h.projection.3.weight") is not None else None
    if yr is None:                       # dropout 0 leaves the Dropout module out: the second Linear is projection.2
        yr = F.normalize(F.linear(F.relu(F.linear(xr, sd["h.projection.0.weight"], sd["h.projection.0.bias"])),
                                  sd["h.projection.2.weight"], sd["h.projection.2.bias"]), p=2, dim=1)
    (yr * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = head(xg)
    assert y.shape == (B, dout)
    close(y, yr, 1e-5, "projection")
    close(y.norm(dim=1), torch.ones(B), 1e-5, "unit rows")
    (y * w.cuda()).sum().backward()
    close(xg.grad, xr.grad, 1e-3, "dx")
    for k, p in head.named_parameters():
        close(p.grad, sd["h." + k].grad, 1e-3, k)
    with pytest.raises(RuntimeError, match="shapes cannot be multiplied"):
        head(torch.randn(B, din + 1).cuda())


def test_projection_head_dropout_mask_is_the_same_forward_and_backward():
    """The hidden layer is exposed by an identity second layer without normalisation: the kept entries are scaled by
    1/(1-p), about a fraction p is dropped, and the backward pass applies the same mask."""
    from ocean.models.components import ProjectionHead
    torch.manual_seed(3)
    B, din, hid, p = 256, 48, 64, 0.25
    head = ProjectionHead(din, hid, hidden_dim=hid, dropout=p).cuda().train()
    head.normalize = False
    lin0, lin1 = head.projection[0], head.projection[3]
    with torch.no_grad():
        lin1.weight.copy_(torch.eye(hid))
        lin1.bias.zero_()
    x = torch.randn(B, din).cuda().requires_grad_(True)
    w = torch.randn(B, hid).cuda()
    y = head(x)
    (y * w).sum().backward()
    with torch.no_grad():
        h = F.relu(F.linear(x, lin0.weight, lin0.bias))
    live = h > 0
    keep = y != 0                        # where the ReLU is live, an entry is zero only if it was dropped
    frac = keep.sum().item() / live.sum().item()
    assert abs(frac - (1 - p)) < 0.02, frac
    close(y[keep], h[keep] / (1 - p), 1e-5, "kept entries")
    mask = keep.float() / (1 - p)
    xr = x.detach().clone().requires_grad_(True)
    hr = F.relu(F.linear(xr, lin0.weight.detach(), lin0.bias.detach())) * mask
    (hr * w).sum().backward()
    close(x.grad, xr.grad, 1e-3, "dx under the mask")
    # a second call draws another mask
    y2 = head(x.detach())
    assert ((y2 != 0) != (y != 0)).any()


def _module(dims, P, K, seed):
    from ocean.models.components import ContrastiveLearningModule
    torch.manual_seed(seed)
    m = ContrastiveLearningModule(dims, projection_dim=P, num_negatives=K)
    for h in m.projection_heads.values():
        for q in h.projection:
            if isinstance(q, torch.nn.Dropout):
                q.p = 0.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.cuda(), sd


@pytest.mark.parametrize("negatives", ["bank", "in_batch"])
@pytest.mark.parametrize("dims,B", [({"temporal": 32, "spatial": 32, "log": 32}, 4), ({"a": 24, "b": 40}, 7),
                                    ({"a": 16, "b": 48, "c": 16, "d": 20}, 3), ({"only": 12}, 5)])
def test_contrastive_module_forward_matches_oracle(dims, B, negatives):
    P, K = 32, 16
    m, sd = _module(dims, P, K, seed=B)
    m.negatives = negatives
    m.train()
    rsd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "memory" not in k else v.clone()) for k, v in sd.items()}
    bank, ptr = rsd["memory_bank"], int(rsd["memory_ptr"])
    for step in range(3):
        torch.manual_seed(100 + step)
        feats = {k: torch.randn(B, d) for k, d in dims.items()}
        feats_with_stranger = dict(feats, unknown_modality=torch.randn(B, 9))      # no head: ignored (graph_fusion.py:219-221)
        for v in rsd.values():
            if v.requires_grad:
                v.grad = None
        proj_r = {k: R.projection_head(rsd, f"projection_heads.{k}.", x) for k, x in feats.items()}
        neg = bank.clone().detach() if negatives == "bank" else None
        losses_r = {f"{a}_{b}_contrastive": R.infonce(proj_r[a], proj_r[b], neg) for a, b in itertools.combinations(dims, 2)}
        if losses_r:
            sum(losses_r.values()).backward()
        m.zero_grad(set_to_none=True)
        out = m({k: v.cuda() for k, v in feats_with_stranger.items()}, return_projections=True)
        assert set(out) == {"losses", "projections"} and list(out["losses"]) == list(losses_r)
        assert list(out["projections"]) == list(dims)
        for k in losses_r:
            close(out["losses"][k], losses_r[k], 1e-3, k)
        for k in dims:
            close(out["projections"][k], proj_r[k], 1e-5, k)
        if losses_r:
            sum(out["losses"].values()).backward()
            for k, p in m.named_parameters():
                close(p.grad, rsd[k].grad, 1e-3, k)
        if negatives == "bank":
            ptr = R.memory_bank_update(bank, ptr, torch.cat([proj_r[k] for k in dims], 0))
        close(m.memory_bank, bank, 1e-5, "memory bank")
        assert int(m.memory_ptr) == ptr
        assert set(m({k: v.cuda() for k, v in feats.items()})) == {"losses"}       # projections only on request
        if negatives == "bank":                                                   # that call wrote the same rows once more
            ptr = R.memory_bank_update(bank, ptr, torch.cat([proj_r[k] for k in dims], 0))
    # eval: losses only, the bank stays
    m.eval()
    before, pbefore = m.memory_bank.clone(), int(m.memory_ptr)
    out = m({k: v.cuda() for k, v in feats.items()})
    assert set(out) == {"losses"}
    assert torch.equal(m.memory_bank, before) and int(m.memory_ptr) == pbefore
