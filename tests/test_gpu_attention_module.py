"""MultiHeadAttention called on its own (the reference exports it, ocean/models/components/__init__.py; forward:
multi_factor_attention.py:108-148 with ScaledDotProductAttention :34-65) through the HIP path against the same
arithmetic in plain torch on the CPU.  fp32; outputs 1e-5, gradients 1e-3."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e}"


def close_param_grads(module, ref_grads, what=""):
    """Every parameter gradient to 1e-3, except the key projection's bias: a constant added to every key changes no
    softmax, so that gradient is zero up to rounding on both sides (and the HIP side accumulates it with atomics)."""
    named = dict(module.named_parameters())
    for name, prm in named.items():
        if name.endswith("w_k.bias"):
            wk = named[name[:-len("bias")] + "weight"].grad
            assert float(prm.grad.abs().max()) <= 1e-4 * max(float(wk.abs().max()), 1e-3), f"{what} {name}"
            continue
        close(prm.grad, ref_grads[name].grad, 1e-3, f"{what} {name}")


def reference_mha(sd, q, k, v, mask, heads, keep_a=None, keep_o=None, p=0.0):
    """multi_factor_attention.py:125-148 and :51-65; keep_a / keep_o: dropout masks (1 = kept) or None."""
    B, L, D = q.shape
    dk = D // heads
    lin = lambda n, x: F.linear(x, sd[n + ".weight"], sd.get(n + ".bias"))
    Q = lin("w_q", q).view(B, L, heads, dk).transpose(1, 2)
    K = lin("w_k", k).view(B, L, heads, dk).transpose(1, 2)
    V = lin("w_v", v).view(B, L, heads, dk).transpose(1, 2)
    scores = Q @ K.transpose(-2, -1) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1).expand(B, heads, L, L) == 0, -1e9)
    w = torch.softmax(scores, dim=-1)
    if keep_a is not None:
        w = w * keep_a / (1 - p)
    out = lin("w_o", (w @ V).transpose(1, 2).contiguous().view(B, L, D))
    if keep_o is not None:
        out = out * keep_o / (1 - p)
    return out, w


@pytest.mark.parametrize("B,L,D,H,bias,masked", [(2, 5, 32, 4, True, False), (3, 1, 128, 8, True, False), (2, 70, 64, 8, True, True),
                                                 (1, 130, 48, 3, False, True), (4, 16, 256, 8, True, False)])
@pytest.mark.parametrize("training", [False, True])
def test_multi_head_attention_matches_the_reference_arithmetic(B, L, D, H, bias, masked, training):
    from ocean.models.components import MultiHeadAttention
    torch.manual_seed(B * 100 + L)
    m = MultiHeadAttention(D, H, dropout=0.0 if training else 0.1, bias=bias)
    m.train(training)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda()
    q, k, v = torch.randn(B, L, D), torch.randn(B, L, D), torch.randn(B, L, D)
    mask = None
    if masked:
        mask = (torch.rand(B, L, L) > 0.3).long()
        mask[:, torch.arange(L), torch.arange(L)] = 1                 # no fully masked row
    wo, wa = torch.randn(B, L, D), torch.randn(B, H, L, L)
    qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
    out_r, att_r = reference_mha(sd, qr, kr, vr, mask, H)
    ((out_r * wo).sum() + (att_r * wa).sum()).backward()
    qg, kg, vg = (x.cuda().requires_grad_(True) for x in (q, k, v))
    out, att = m(qg, kg, vg, None if mask is None else mask.cuda())
    assert out.shape == (B, L, D) and att.shape == (B, H, L, L)
    close(out, out_r, 1e-5, "output")
    close(att, att_r, 1e-5, "attention weights")
    ((out * wo.cuda()).sum() + (att * wa.cuda()).sum()).backward()
    for name, a, b in (("dq", qg, qr), ("dk", kg, kr), ("dv", vg, vr)):
        close(a.grad, b.grad, 1e-3, name)
    close_param_grads(m, sd)
    # gradient through one output alone
    for which in (0, 1):
        qg2 = q.cuda().requires_grad_(True)
        o2 = m(qg2, kg.detach(), vg.detach(), None if mask is None else mask.cuda())
        qr2 = q.clone().requires_grad_(True)
        r2 = reference_mha({n: t.detach() for n, t in sd.items()}, qr2, k, v, mask, H)
        (o2[which] * (wo.cuda() if which == 0 else wa.cuda())).sum().backward()
        (r2[which] * (wo if which == 0 else wa)).sum().backward()
        close(qg2.grad, qr2.grad, 1e-3, f"dq through output {which}")


def test_multi_head_attention_dropout_masks_are_shared_by_forward_and_backward():
    from ocean.models.components import MultiHeadAttention
    torch.manual_seed(5)
    B, L, D, H, p = 8, 12, 64, 4, 0.25
    m = MultiHeadAttention(D, H, dropout=p).cuda().train()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    q, k, v = torch.randn(B, L, D), torch.randn(B, L, D), torch.randn(B, L, D)
    qg = q.cuda().requires_grad_(True)
    out, att = m(qg, k.cuda(), v.cuda())
    keep_a, keep_o = (att != 0).float().cpu(), (out != 0).float().cpu()
    assert abs(keep_a.mean().item() - (1 - p)) < 0.03 and abs(keep_o.mean().item() - (1 - p)) < 0.03
    qr = q.clone().requires_grad_(True)
    out_r, att_r = reference_mha(sd, qr, k, v, None, H, keep_a, keep_o, p)
    close(out, out_r, 1e-5, "output under the masks")
    close(att, att_r, 1e-5, "weights under the mask")
    wo = torch.randn(B, L, D)
    (out * wo.cuda()).sum().backward()
    (out_r * wo).sum().backward()
    close(qg.grad, qr.grad, 1e-3, "dq")
    close_param_grads(m, sd)
    out2, _ = m(q.cuda(), k.cuda(), v.cuda())
    assert ((out2 != 0) != (out != 0)).any()            # another call, another mask


def test_multi_head_attention_rejects_what_the_reference_rejects():
    from ocean.models.components import MultiHeadAttention
    m = MultiHeadAttention(32, 4).cuda().eval()
    x = torch.randn(2, 5, 32).cuda()
    with pytest.raises(RuntimeError):
        m(x, torch.randn(2, 6, 32).cuda(), x)           # the reference's .view with the query's length fails
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 5, 16).cuda(), x, x)


@pytest.mark.parametrize("B,L,D,H,masked", [(2, 6, 32, 4, False), (3, 40, 128, 8, True), (1, 1, 64, 8, False)])
@pytest.mark.parametrize("training", [False, True])
def test_cross_modal_attention_matches_the_reference_arithmetic(B, L, D, H, masked, training):
    """CrossModalAttention.forward (multi_factor_attention.py:190-215): attention, residual + LayerNorm, feed-forward,
    residual + LayerNorm."""
    from ocean.models.components import CrossModalAttention
    torch.manual_seed(B * 10 + L)
    m = CrossModalAttention(D, H, dropout=0.0 if training else 0.1)
    m.train(training)
    for q_ in m.modules():                         # eval ignores dropout; training runs with p = 0
        if isinstance(q_, torch.nn.Dropout) and training:
            q_.p = 0.0
    with torch.no_grad():                          # LayerNorm parameters away from their (1, 0) initial values
        for ln in (m.layer_norm, m.ff_layer_norm):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.uniform_(-0.3, 0.3)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda()
    x, y = torch.randn(B, L, D), torch.randn(B, L, D)
    mask = None
    if masked:
        mask = (torch.rand(B, L, L) > 0.3).long()
        mask[:, torch.arange(L), torch.arange(L)] = 1
    w = torch.randn(B, L, D)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    att_sd = {k[len("multihead_attention."):]: v for k, v in sd.items() if k.startswith("multihead_attention.")}
    a_r, w_r = reference_mha(att_sd, xr, yr, yr, mask, H)
    y1 = F.layer_norm(xr + a_r, (D,), sd["layer_norm.weight"], sd["layer_norm.bias"])
    ffo = F.linear(F.relu(F.linear(y1, sd["feed_forward.0.weight"], sd["feed_forward.0.bias"])), sd["feed_forward.3.weight"],
                   sd["feed_forward.3.bias"])
    out_r = F.layer_norm(y1 + ffo, (D,), sd["ff_layer_norm.weight"], sd["ff_layer_norm.bias"])
    (out_r * w).sum().backward()
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    out, wts = m(xg, yg, None if mask is None else mask.cuda())
    close(out, out_r, 1e-5, "output")
    close(wts, w_r, 1e-5, "attention weights")
    (out * w.cuda()).sum().backward()
    close(xg.grad, xr.grad, 1e-3, "d query modality")
    close(yg.grad, yr.grad, 1e-3, "d key/value modality")
    close_param_grads(m, sd)


def test_cross_modal_attention_trains_with_dropout():
    from ocean.models.components import CrossModalAttention
    torch.manual_seed(2)
    m = CrossModalAttention(64, 8, dropout=0.2).cuda().train()
    x, y = torch.randn(4, 9, 64).cuda().requires_grad_(True), torch.randn(4, 9, 64).cuda()
    out, wts = m(x, y)
    out2, _ = m(x, y)
    assert bool(torch.isfinite(out).all()) and not torch.equal(out, out2)        # masks differ call to call
    assert abs((wts == 0).float().mean().item() - 0.2) < 0.05
    out.square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0


@pytest.mark.parametrize("shape,masked", [((3, 7, 16), False), ((2, 4, 9, 8), True), ((1, 1, 70, 32), False), ((2, 5, 24), True)])
@pytest.mark.parametrize("training", [False, True])
def test_scaled_dot_product_attention_alone(shape, masked, training):
    """ScaledDotProductAttention.forward (multi_factor_attention.py:34-65) on [B, L, d] and [B, H, L, d] inputs; the
    scale is 1/sqrt(d_model) of the constructor, whatever the inputs' last dimension."""
    from ocean.models.components import ScaledDotProductAttention
    torch.manual_seed(sum(shape))
    d_model = 11
    m = ScaledDotProductAttention(d_model, dropout=0.0 if training else 0.2).cuda().train(training)
    q, k, v = (torch.randn(*shape) for _ in range(3))
    L = shape[-2]
    mask = None
    if masked:
        mask = (torch.rand(shape[0], *((1,) if len(shape) == 4 else ()), L, L) > 0.3).long()
        mask[..., torch.arange(L), torch.arange(L)] = 1
    qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
    scores = qr @ kr.transpose(-2, -1) / math.sqrt(d_model)
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e9)
    w_r = torch.softmax(scores, dim=-1)
    o_r = w_r @ vr
    wo, ww = torch.randn_like(o_r), torch.randn_like(w_r)
    ((o_r * wo).sum() + (w_r * ww).sum()).backward()
    qg, kg, vg = (x.cuda().requires_grad_(True) for x in (q, k, v))
    o, w = m(qg, kg, vg, None if mask is None else mask.cuda())
    close(o, o_r, 1e-5, "attended values")
    close(w, w_r, 1e-5, "weights")
    ((o * wo.cuda()).sum() + (w * ww.cuda()).sum()).backward()
    for name, a, b in (("dq", qg, qr), ("dk", kg, kr), ("dv", vg, vr)):
        close(a.grad, b.grad, 1e-3, name)


def test_multi_head_attention_random_shapes():
    """Seeded random batch sizes, lengths (around the 64-lane boundaries too), head counts and head widths (odd ones
    included), with and without masks, eval mode: output, weights and every gradient."""
    import random
    from ocean.models.components import MultiHeadAttention
    rnd = random.Random(7)
    for case in range(16):
        H = rnd.choice([1, 2, 3, 4, 8])
        dk = rnd.choice([1, 3, 7, 8, 16, 33, 64, 70])
        D, B = H * dk, rnd.choice([1, 2, 5])
        L = rnd.choice([1, 2, 31, 63, 64, 65, 100, 129])
        torch.manual_seed(case)
        m = MultiHeadAttention(D, H, dropout=0.1, bias=rnd.random() < 0.8).eval()
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        m = m.cuda()
        q, k, v = (torch.randn(B, L, D) for _ in range(3))
        mask = None
        if rnd.random() < 0.5:
            mask = (torch.rand(B, L, L) > 0.4).long()
            mask[:, torch.arange(L), torch.arange(L)] = 1
        wo, wa = torch.randn(B, L, D), torch.randn(B, H, L, L)
        qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
        out_r, att_r = reference_mha(sd, qr, kr, vr, mask, H)
        ((out_r * wo).sum() + (att_r * wa).sum()).backward()
        qg, kg, vg = (x.cuda().requires_grad_(True) for x in (q, k, v))
        out, att = m(qg, kg, vg, None if mask is None else mask.cuda())
        what = f"case {case}: B={B} L={L} H={H} dk={dk} mask={mask is not None}"
        close(out, out_r, 1e-5, what + " output")
        close(att, att_r, 1e-5, what + " weights")
        ((out * wo.cuda()).sum() + (att * wa.cuda()).sum()).backward()
        for name, a, b in (("dq", qg, qr), ("dk", kg, kr), ("dv", vg, vr)):
            close(a.grad, b.grad, 1e-3, what + " " + name)
        close_param_grads(m, sd, what)
