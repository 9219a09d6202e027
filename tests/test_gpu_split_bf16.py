"""The opt-in split-bf16 arithmetic of the large convolution products (ocean.set_split_bf16): every fp32 operand is split
into two bf16 values and the product formed by three bf16 matrix instructions with fp32 accumulation.  The reference has
no such mode; these tests bound what it costs in accuracy: each product to ~1e-5 of the operands' scale against an fp64
result (the fp32 kernels: ~1e-6 and below), and the whole update step at the headline shape still inside the bounds the
fp32 path is held to against the oracle (probabilities 1e-4 relative, identical top-k, losses / gradient norm /
gradients 1e-3)."""
import pytest
import torch

from oracle import ocean_ref as R
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["all", "backward"])
def split_mode(request):
    import ocean
    prev = ocean.set_split_bf16(request.param)
    try:
        yield request.param
    finally:
        ocean.set_split_bf16(prev)


def _relerr(got, ref):
    ref = ref.double()
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item()


def _conv_products(T=128, B=256, C=256, d=4):
    """Forward convolution [B*T, C] x 3 taps and its weight gradient through the C ABI; fp64 results beside them."""
    from ocean import _hip
    L = _hip.lib()
    M = B * T
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, C, generator=g)
    w = torch.randn(3, C, C, generator=g) * 0.05
    dy = torch.randn(M, C, generator=g)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    y = torch.empty(M, C, device="cuda")
    dw = torch.zeros(3, C, C, device="cuda")
    _hip.check(L.ocean_conv_gemm_f32(xd.data_ptr(), C, wd.data_ptr(), C, y.data_ptr(), C, M, C, C, 0, 0, 3, T, -d, d, C * C, 0, 1,
                                     _hip.stream()), "conv forward")
    _hip.check(L.ocean_conv_gemm_f32(dyd.data_ptr(), C, xd.data_ptr(), C, dw.data_ptr(), C, C, C, M, 1, 1, 3, T, -d, d, 0, C * C, 0,
                                     _hip.stream()), "conv weight gradient")
    torch.cuda.synchronize()
    x3 = x.double().view(B, T, C)
    yref = torch.zeros(B, T, C, dtype=torch.float64)
    dwref = torch.zeros(3, C, C, dtype=torch.float64)
    dy3 = dy.double().view(B, T, C)
    for t in range(3):
        s = (t - 1) * d                              # tap t reads time index i + s, zero outside [0, T)
        shifted = torch.zeros_like(x3)
        if s >= 0:
            shifted[:, :T - s] = x3[:, s:]
        else:
            shifted[:, -s:] = x3[:, :T + s]
        yref += shifted @ w[t].double().t()
        dwref[t] = dy3.reshape(M, C).t() @ shifted.reshape(M, C)
    return _relerr(y, yref.view(M, C)), _relerr(dw, dwref)


def test_switch_returns_previous_setting_and_changes_the_products():
    import ocean
    assert ocean.split_bf16() == 0                  # off unless asked for
    e_fwd, e_wg = _conv_products()
    prev = ocean.set_split_bf16(True)
    try:
        assert prev == 0 and ocean.split_bf16() == 1
        s_fwd, s_wg = _conv_products()
        assert ocean.set_split_bf16("backward") == 1 and ocean.split_bf16() == 2
        b_fwd, b_wg = _conv_products()               # called outside a backward driver: fp32 products
        with pytest.raises(ValueError):
            ocean.set_split_bf16("sometimes")
    finally:
        assert ocean.set_split_bf16(False) == 2
    assert ocean.split_bf16() == 0
    assert b_fwd < 2e-6 and b_wg < 2e-6, (b_fwd, b_wg)
    assert e_fwd < 2e-6 and e_wg < 2e-6, (e_fwd, e_wg)
    # three bf16 products carry about 16 bits: worse than fp32, far better than one bf16 product (4e-3)
    assert e_fwd < s_fwd < 3e-5, (e_fwd, s_fwd)
    assert e_wg < s_wg < 3e-5, (e_wg, s_wg)


def test_headline_shape_step_in_split_mode_stays_inside_the_fp32_bounds(split_mode):
    """Batch 1024 x 128 timesteps x 256 services (every convolution product on the split kernels, or those of the backward
    pass only), dropout 0, against the CPU oracle: the same bounds as the fp32 path's full-size test."""
    import ocean
    from ocean.data import ServiceGraph
    from ocean.optim import FusedAdam
    N, B, T = 256, 1024, 128
    cfg = ocean.default_config()
    cfg.model.num_services = N
    m = ocean.OCEANModel(cfg)
    for q in m.modules():
        if isinstance(q, torch.nn.Dropout):
            q.p = 0.0
        if isinstance(q, torch.nn.MultiheadAttention) or q.__class__.__name__ == "GATConv":
            q.dropout = 0.0
    sd = synth.fill_state_dict(m.state_dict(), 47)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.contrastive_negatives = "in_batch"
    gen = torch.Generator().manual_seed(19)
    adj, nf = synth.random_digraph(N, 8.0, 17), torch.randn(N, 12, generator=gen)
    metrics, logs = torch.randn(B, T, 12, generator=gen), torch.randn(B, 768, generator=gen)
    labels = torch.zeros(B, N)
    labels[torch.arange(B), torch.randint(0, N, (B,), generator=gen)] = 1.0
    out = m(metrics, ServiceGraph(adj, nf), logs)
    losses = m.compute_loss(out, labels.cuda())
    losses["total_loss"].backward()
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    opt.step(max_norm=1.0, write_clipped=False)
    gnorm = float(opt.grad_norm())
    got = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    probs = out["root_cause_probs"].detach().cpu()
    host_losses = {k: float(losses[k].detach()) for k in ("total_loss", "prediction_loss", "contrastive_loss")}
    del out, losses
    torch.cuda.empty_cache()
    rsd = R.require_grad(sd)
    ro = R.model_forward(rsd, metrics, adj, nf, logs, training=True, negatives="in_batch")
    rl = R.compute_loss(ro, labels)
    rl["total_loss"].backward()
    rprobs = ro["root_cause_probs"].detach()
    assert ((probs - rprobs).abs() / rprobs).max().item() < 1e-4
    for k in (1, 3, 5, 10):
        assert torch.equal(torch.topk(probs, k, dim=-1).indices, torch.topk(rprobs, k, dim=-1).indices), k
    for k, v in host_losses.items():
        assert abs(v - float(rl[k])) <= 1e-3 * abs(float(rl[k])), (k, v, float(rl[k]))
    tk = [k for k in R.trainable_keys(rsd) if rsd[k].grad is not None]
    rnorm = torch.sqrt(sum((rsd[k].grad.double() ** 2).sum() for k in tk)).item()
    assert abs(gnorm - rnorm) <= 1e-3 * rnorm, (gnorm, rnorm)
    for k in tk:
        ref = rsd[k].grad.double()
        err = (got[k].double() - ref).abs().max().item()
        assert err <= 1e-3 * max(ref.abs().max().item(), 1e-3), (k, err)


def test_a_learner_records_a_new_step_when_the_mode_changes():
    """Recorded steps are keyed on the arithmetic mode: switching it makes the learner record (and then replay) a second
    step with the other kernels instead of replaying the first one."""
    import ocean
    from ocean.data import ServiceGraph
    from ocean.models.training import OnlineLearner
    N, B, T = 256, 256, 128                          # 256 row tiles: the large kernels are the ones in use
    cfg = ocean.default_config()
    cfg.model.num_services = N
    torch.manual_seed(1)
    m = ocean.OCEANModel(cfg).cuda()
    m.contrastive_negatives = "in_batch"
    learner = OnlineLearner(m, cfg)
    learner.use_hip_graph = True
    gen = torch.Generator().manual_seed(2)
    sg = ServiceGraph(synth.random_digraph(N, 8.0, 3), torch.randn(N, 12, generator=gen))
    x, l = torch.randn(B, T, 12, generator=gen).cuda(), torch.randn(B, 768, generator=gen).cuda()
    y = torch.zeros(B, N)
    y[torch.arange(B), torch.randint(0, N, (B,), generator=gen)] = 1.0
    y = y.cuda()
    for _ in range(4):
        learner.process_tensors(x, sg, l, y)
    assert len(learner._graphed) == 1
    prev = ocean.set_split_bf16(True)
    try:
        for _ in range(4):
            r = learner.process_tensors(x, sg, l, y)
        assert len(learner._graphed) == 2
        assert r["batch_loss"] == r["batch_loss"] and r["batch_loss"] > 0          # finite
    finally:
        ocean.set_split_bf16(prev)
    learner.process_tensors(x, sg, l, y)
    assert len(learner._graphed) == 2                # back on the first recording


def test_large_row_dcnn_gradients_keep_their_bounds_with_the_backward_products_split():
    """The two random-shape DilatedCNN cases at >= 16 384 rows whose element-wise gradient bound (5 % of the largest
    element) the all-products setting exceeds - more ReLU inputs change sign in a 16-bit forward pass - run unchanged,
    bounds included, when only the backward products are split."""
    import importlib.util
    import pathlib
    import ocean
    spec = importlib.util.spec_from_file_location("_dcnn_parity_tests", pathlib.Path(__file__).with_name("test_gpu_dcnn.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    prev = ocean.set_split_bf16("backward")
    try:
        mod.test_against_oracle_random(256, 128, [64, 128], [3, 32], 3)
        mod.test_against_oracle_random(512, 64, [64, 128], [2, 5], 3)
    finally:
        ocean.set_split_bf16(prev)
