"""Whole-model parity on the GPU: OCEANModel / OnlineLearner through the HIP
path against fixtures recorded from the reference's own classes
(oracle/gen_golden.py).  fp32, dropout 0.  Tolerances are the north-star
ones: root_cause_probs 1e-4 relative, losses and gradients 1e-3, identical
top-k rankings."""
import ast
from datetime import datetime

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(got, ref, tol):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"max err {err:.3e} scale {scale:.3e}"


def close_elem(got, ref, rel, floor, what=""):
    """Every element on its own: |got - ref| <= rel * |ref| + floor (floor is an absolute amount)."""
    got, ref = torch.as_tensor(got).detach().double().cpu().flatten(), torch.as_tensor(ref).detach().double().cpu().flatten()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if not ref.numel():
        return
    excess = (got - ref).abs() - (rel * ref.abs() + floor)
    i = int(excess.argmax())
    assert excess[i] <= 0, f"{what}: element {i}: got {got[i]:.6e} ref {ref[i]:.6e} (allowed {rel * abs(ref[i]) + floor:.3e})"


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if m.__class__.__name__ == "GATConv":
            m.dropout = 0.0


def tiny_config():
    import ocean
    cfg = ocean.OCEANConfig()
    cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
    cfg.model.hidden_dim = cfg.model.attention_dim = 16
    cfg.model.attention_num_heads, cfg.model.gnn_num_heads = 4, 2
    cfg.model.fusion_projection_dim, cfg.model.num_services = 32, 6
    cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = 5, 24, 7
    return cfg


def build(cfg, sd):
    import ocean
    m = ocean.OCEANModel(cfg)
    zero_dropout(m)
    m.load_state_dict(sd)
    return m.cuda()


def graph_of(g):
    from ocean.data import ServiceGraph
    return ServiceGraph(adjacency_matrix=T(g["adj"]), node_features=T(g["node_features"]))


def test_readme_example_eval_and_topk(golden):
    import ocean
    from ocean.data import ServiceGraph
    g = golden("model_readme")
    cfg = ocean.default_config()
    m = ocean.OCEANModel(cfg)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), int(g["meta_seed"])))
    m = m.cuda().eval()
    sg = ServiceGraph(adjacency_matrix=T(g["adj"]), node_features=T(g["node_features"]),
                      service_names=[f"service_{i}" for i in range(5)])
    with torch.no_grad():
        o = m(T(g["metrics"]), sg, T(g["logs"]), return_intermediate=True)
    it = o["intermediate"]
    for k, name in (("temporal_features", "temporal"), ("spatial_features", "spatial"),
                    ("attended_features", "attended"), ("fused_representation", "fused")):
        close(it[k], g[name], 1e-4)
    close(o["root_cause_logits"], g["logits"], 1e-4)
    rel = ((o["root_cause_probs"].cpu() - T(g["probs"])).abs() / T(g["probs"])).max().item()
    assert rel < 1e-4
    assert torch.equal(torch.topk(o["root_cause_probs"], 10, dim=-1).indices.cpu(), T(g["top10"]))
    r = m.predict_root_cause(T(g["metrics"]), ServiceGraph(T(g["adj"]), T(g["node_features"])), T(g["logs"]), top_k=3)
    assert torch.equal(r["top_k_indices"].cpu(), T(g["top10"])[:, :3])
    # head is wider (100) than the graph (5 names): the reference raises IndexError here too
    with pytest.raises(IndexError):
        m.predict_root_cause(T(g["metrics"]), sg, T(g["logs"]), top_k=10)


def run_steps(g, m, neg, tiny):
    from ocean.optim import FusedAdam
    m.contrastive_negatives = neg
    sg = graph_of(g)
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    names = dict(m.named_parameters())
    for step in range(3):
        pre = f"s{step}/"
        m.train()
        out = m(T(g[pre + "metrics"]), sg, T(g[pre + "logs"]), return_intermediate=True)
        losses = m.compute_loss(out, T(g[pre + "labels"]).cuda())
        rel = ((out["root_cause_probs"].detach().cpu() - T(g[pre + "probs"])).abs() / T(g[pre + "probs"])).max().item()
        assert rel < 1e-4, (step, rel)
        it = out["intermediate"]
        for k, name in (("temporal_features", "temporal"), ("spatial_features", "spatial"),
                        ("attended_features", "attended"), ("fused_representation", "fused")):
            close(it[k], g[pre + name], 2e-4)
        for k in g.files:
            if k.startswith(pre + "loss/"):
                close(losses[k[len(pre) + 5:]], g[k], 1e-3)
        opt.zero_grad()
        losses["total_loss"].backward()
        if step == 0:
            # same None-gradient set as the reference (52 tensors in the default model); the 12 dead
            # w_q / w_k tensors carry exact zeros, as in the reference
            assert {k for k, p in names.items() if p.grad is None} == set(str(k) for k in g["none_grad"])
            dead_qk = [k for k in names if (".w_q." in k or ".w_k." in k) and "_self_attention" in k]
            assert len(dead_qk) == 12 and all(float(names[k].grad.abs().max()) == 0.0 for k in dead_qk)
        opt.step(max_norm=1.0)
        torch.cuda.synchronize()
        norm = float(opt.grad_norm())
        assert abs(norm - float(g[pre + "grad_norm"])) <= 1e-3 * float(g[pre + "grad_norm"])
        if step == 0:
            for k, p in names.items():
                if p.grad is None:
                    continue
                if tiny:
                    close(p.grad, g["s0/cgrad/" + k], 1e-3)
                else:
                    close(synth.digest(p.grad.cpu()), g["s0/cgdig/" + k], 1e-3)
                    # 256 evenly spaced elements of every gradient, each on its own: 1e-3 of the element plus 1e-5 of
                    # the tensor's largest sampled magnitude (sums of ~1e3 fp32 terms in another order)
                    ref = T(g["s0/cgsamp/" + k])
                    close_elem(synth.samples(p.grad.cpu()), ref, 1e-3, 1e-5 * max(float(ref.abs().max()), 1e-6), "grad " + k)
        cm = m.fusion_module.contrastive_module
        close(cm.memory_bank, g[pre + "bank"], 1e-4)
        assert int(cm.memory_ptr) == int(g[pre + "ptr"][0])
        if step in (0, 2):
            for k, v in m.state_dict().items():
                if tiny:
                    close(v, g[pre + "sd_after/" + k], 5e-4)
                else:
                    close(synth.digest(v.cpu()), g[pre + "sddig_after/" + k], 2e-3)
                    if v.is_floating_point():
                        # sampled elements of the state after the step.  Adam moves a weight by lr * m / (sqrt(v) + eps):
                        # where the gradient is rounding noise around zero that quotient can take either sign, so
                        # an element may be off by up to 2 * lr per step taken; everything else agrees to 1e-3
                        close_elem(synth.samples(v.cpu()), T(g[pre + "sdsamp_after/" + k]), 1e-3, 2e-3 * (step + 1), "state " + k)


@pytest.mark.parametrize("neg", ["bank", "in_batch"])
def test_three_update_steps_tiny(golden, neg):
    g = golden(f"model_train_tiny_{neg}")
    sd = {k[3:]: T(g[k]).clone() for k in g.files if k.startswith("sd/")}
    run_steps(g, build(tiny_config(), sd), neg, tiny=True)


@pytest.mark.parametrize("neg", ["bank", "in_batch"])
def test_three_update_steps_default_arch(golden, neg):
    import ocean
    g = golden(f"model_train_default_{neg}")
    cfg = ocean.default_config()
    m0 = ocean.OCEANModel(cfg)
    run_steps(g, build(cfg, synth.fill_state_dict(m0.state_dict(), int(g["meta_seed"]))), neg, tiny=False)


def test_online_learner_return_dicts(golden):
    import ocean
    from ocean.data import DatasetSample, ServiceGraph
    from ocean.models.training import OnlineLearner
    g = golden("online_learner")
    cfg = tiny_config()
    m0 = ocean.OCEANModel(cfg)
    m = build(cfg, synth.fill_state_dict(m0.state_dict(), int(g["meta_seed"])))
    learner = OnlineLearner(m, cfg)
    sg = ServiceGraph(adjacency_matrix=T(g["adj"]), node_features=T(g["node_features"]))
    samples = [DatasetSample(metrics=T(g[f"in/metrics{i}"]), logs=T(g[f"in/logs{i}"]), graph=sg, label=T(g[f"in/label{i}"]),
                             timestamp=datetime(2024, 1, 1)) for i in range(6)]
    for i in range(3):
        r = learner.process_sample(samples[i])
        for k, v in r.items():
            ref = g[f"sample{i}/{k}"]
            assert abs(float(v) - float(ref)) <= 1e-3 * max(abs(float(ref)), 1e-3), (i, k, v, ref)
    r = learner.process_batch(samples[3:6])
    for k, v in r.items():
        ref = g[f"batch/{k}"]
        assert abs(float(v) - float(ref)) <= 1e-3 * max(abs(float(ref)), 1e-3), (k, v, ref)


def test_model_requires_gpu_and_library():
    import ocean
    from ocean import _hip
    from ocean.data import ServiceGraph
    m = ocean.OCEANModel(tiny_config())
    with pytest.raises(_hip.OceanHipError):
        m(torch.randn(1, 4, 5), ServiceGraph(torch.eye(3), torch.randn(3, 7)), torch.randn(1, 24))


@pytest.mark.parametrize("neg", ["bank", "in_batch"])
def test_dropout_backward_matches_finite_differences(golden, neg):
    """All dropout sites active (reference defaults): under a fixed CPU seed every component draws the same mask
    seeds, so the training-mode loss is a deterministic function of the parameters; central differences along one
    parameter tensor at a time must agree with the analytic gradient, which requires every backward kernel to
    regenerate its forward mask (attention, fusion, contrastive projections, head, GAT, dilated CNN)."""
    import ocean
    g = golden(f"model_train_tiny_{neg}")
    cfg = tiny_config()
    m = ocean.OCEANModel(cfg)
    m.load_state_dict({k[3:]: T(g[k]).clone() for k in g.files if k.startswith("sd/")})
    m = m.cuda().train()
    m.contrastive_negatives = neg
    sg = graph_of(g)
    metrics, logs, labels = T(g["s0/metrics"]).cuda(), T(g["s0/logs"]).cuda(), T(g["s0/labels"]).cuda()
    cm = m.fusion_module.contrastive_module
    bank0, ptr0 = cm.memory_bank.clone(), cm.memory_ptr.clone()
    running = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}

    def f():
        with torch.no_grad():                       # the forward mutates the bank and the BatchNorm running statistics
            cm.memory_bank.copy_(bank0); cm.memory_ptr.copy_(ptr0)
            for k, v in m.state_dict().items():
                if k in running:
                    v.copy_(running[k])
        torch.manual_seed(1234)
        return m.compute_loss(m(metrics, sg, logs), labels)["total_loss"].double()

    for p in m.parameters():
        p.grad = None
    f().backward()
    names = dict(m.named_parameters())
    picks = ["temporal_encoder.conv_layers.0.conv.weight", "temporal_encoder.conv_layers.1.batch_norm.weight",
             "spatial_encoder.conv_layers.0.lin.weight", "spatial_encoder.conv_layers.1.att_src",
             "attention_module.temporal_projection.weight", "attention_module.log_self_attention.w_v.weight",
             "attention_module.spatial_self_attention.w_o.bias", "attention_module.layer_norms.temporal.weight",
             "attention_module.output_projection.0.weight", "fusion_module.log_processor.0.weight",
             "fusion_module.spatial_processor.3.weight", "fusion_module.attention_fusion.in_proj_weight",
             "fusion_module.contrastive_module.projection_heads.log.projection.0.weight",
             "fusion_module.contrastive_module.projection_heads.temporal.projection.3.weight",
             "fusion_module.output_projection.3.weight", "prediction_head.3.weight"]
    picks = [k for k in picks if k in names and names[k].grad is not None]
    assert len(picks) >= 14, picks
    grads = {k: names[k].grad.clone() for k in picks}
    eps = 2e-3
    for k in picks:
        p = names[k]
        d = torch.randn(p.shape, generator=torch.Generator().manual_seed(5)).cuda() * p.detach().abs().mean().clamp_min(1e-3)
        analytic = (grads[k].double() * d.double()).sum().item()
        with torch.no_grad():
            p.add_(d, alpha=eps)
            fp = f().item()
            p.add_(d, alpha=-2 * eps)
            fm = f().item()
            p.add_(d, alpha=eps)
        numeric = (fp - fm) / (2 * eps)
        assert abs(numeric - analytic) <= 6e-2 * abs(analytic) + 2e-3, (k, numeric, analytic)


@pytest.mark.parametrize("B,E", [(12, 32), (130, 256), (5, 40), (9, 64), (1027, 128), (2, 32)])   # batches that are not
def test_mutual_information_estimates_match_the_per_modality_formula(B, E):      # multiples of 4: blocks straddle modalities
    """The three MINE estimates come from one fused call; each must equal the reference's per-modality expression
    mean(f(cat(out, m))) - log(mean(exp(f(cat(out, m[perm])))) + 1e-8) (graph_fusion.py:475-501) under the same
    three permutations."""
    from ocean.models.components import GraphFusionModule
    torch.manual_seed(5)
    fm = GraphFusionModule(16, 16, 16, fusion_dim=E, projection_dim=16).cuda().train()
    fused = torch.randn(B, E, device="cuda")
    X = torch.randn(B, 3, E, device="cuda")
    perms = GraphFusionModule.draw_permutations(B, "cuda")
    assert perms.shape == (3, B) and all(sorted(p.tolist()) == list(range(B)) for p in perms)
    got = fm._mutual_information(fused, X, perms)
    with torch.no_grad():
        for i, k in enumerate(("temporal", "spatial", "log")):
            m = X[:, i]
            joint = fm.mi_estimator(torch.cat([fused, m], 1))
            marg = fm.mi_estimator(torch.cat([fused, m[perms[i]]], 1))
            ref = joint.mean() - torch.log(torch.exp(marg).mean() + 1e-8)
            assert abs(got[f"{k}_mi"].item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item())), k
    # without explicit permutations a fresh draw is used each call
    a, b = fm._mutual_information(fused, X), fm._mutual_information(fused, X)
    assert all(torch.isfinite(v) for v in a.values()) and any(a[k].item() != b[k].item() for k in a)


@pytest.mark.parametrize("E", [64, 128, 256, 40])
def test_fusion_token_attention_all_widths(E):
    """GraphFusionModule (graph_fusion.py:388-473, nn.MultiheadAttention with 8 heads over the 3 modality tokens)
    forward and input / parameter gradients against the oracle at the widths that take the wave-per-sample attention
    kernels (64, 128, 256) and one that takes the thread-per-head kernel (40)."""
    from ocean.models.components import GraphFusionModule
    from oracle import ocean_ref as R
    torch.manual_seed(5)
    B = 7
    m = GraphFusionModule(12, 10, 14, fusion_dim=E, projection_dim=16, dropout=0.0).eval()
    sd = {"f." + k: v.detach().clone() for k, v in m.state_dict().items()}
    t, s, l = torch.randn(B, 12), torch.randn(B, 10), torch.randn(B, 14)
    w = torch.randn(B, E)
    ref_in = [x.clone().requires_grad_(True) for x in (t, s, l)]
    ref_sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    ref = R.fusion_forward(ref_sd, "f.", *ref_in, training=False, num_heads=8)["fused_representation"]
    (ref * w).sum().backward()
    m = m.cuda()
    got_in = [x.clone().cuda().requires_grad_(True) for x in (t, s, l)]
    got = m(*got_in, compute_contrastive_loss=False)["fused_representation"]
    (got * w.cuda()).sum().backward()
    close(got, ref, 1e-4)
    for a, b in zip(got_in, ref_in):
        close(a.grad, b.grad, 1e-3)
    for name in ("attention_fusion.in_proj_weight", "attention_fusion.in_proj_bias", "attention_fusion.out_proj.weight",
                 "temporal_processor.0.weight", "output_projection.3.bias"):
        close(dict(m.named_parameters())[name].grad, ref_sd["f." + name].grad, 1e-3)


@pytest.mark.parametrize("B,P", [(5, 16), (64, 128), (130, 24)])
def test_three_pair_infonce_equals_three_single_calls_and_the_oracle(B, P):
    """InfoNCELoss.forward on the pairs (temporal, spatial), (temporal, log), (spatial, log) (graph_fusion.py:112-162,
    231-249): the batched entry point against the oracle and against the per-pair entry point, values and gradients,
    with unequal weights on the three losses."""
    from ocean import ops
    from oracle import ocean_ref as R
    g = torch.Generator().manual_seed(B)
    Z = torch.randn(3, B, P, generator=g) * torch.rand(3, B, 1, generator=g).add(0.5)      # not unit rows: the op re-normalises
    w = torch.tensor([1.0, 0.3, -2.0])
    zr = Z.clone().requires_grad_(True)
    ref = torch.stack([R.infonce(zr[a], zr[b], None, 0.1) for a, b in ((0, 1), (0, 2), (1, 2))])
    (ref * w).sum().backward()
    zg = Z.clone().cuda().requires_grad_(True)
    got = ops.infonce_pairs(zg, 0.1)
    (got * w.cuda()).sum().backward()
    close(got, ref, 1e-5)
    close(zg.grad, zr.grad, 1e-3)
    zs = Z.clone().cuda().requires_grad_(True)
    single = torch.stack([ops.infonce(zs[a], zs[b], None, 0.1) for a, b in ((0, 1), (0, 2), (1, 2))])
    (single * w.cuda()).sum().backward()
    close(got, single, 1e-6)
    close(zg.grad, zs.grad, 1e-4)


def test_three_pair_infonce_with_gathered_positives_equals_per_pair_calls():
    """Data-parallel form: anchors are this rank's rows, positives the rows gathered from every rank, labels offset by
    the rank's row base (InfoNCELoss's in-batch branch, graph_fusion.py:144-157, over local rows x global columns)."""
    from ocean import ops
    B, Ball, P, off = 6, 17, 24, 8
    g = torch.Generator().manual_seed(3)
    G = torch.randn(3, Ball, P, generator=g)
    Z = G[:, off:off + B].clone()
    w = torch.tensor([0.7, -1.1, 2.0])
    pairs = ((0, 1), (0, 2), (1, 2))

    def oracle_loss(a, allb):                     # cross entropy of a . allb^T / tau against labels off + r
        a, allb = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(allb, dim=1)
        return torch.nn.functional.cross_entropy(a @ allb.t() / 0.1, torch.arange(B) + off)

    zr, gr = Z.clone().requires_grad_(True), G.clone().requires_grad_(True)
    ref = torch.stack([oracle_loss(zr[a], gr[b]) for a, b in pairs])
    (ref * w).sum().backward()
    zg, gg = Z.clone().cuda().requires_grad_(True), G.clone().cuda().requires_grad_(True)
    got = ops.infonce_pairs(zg, 0.1, gathered=gg, label_offset=off)
    (got * w.cuda()).sum().backward()
    close(got, ref, 1e-5)
    close(zg.grad, zr.grad, 1e-3)
    close(gg.grad, gr.grad, 1e-3)
    zs, gs = Z.clone().cuda().requires_grad_(True), G.clone().cuda().requires_grad_(True)
    single = torch.stack([ops.infonce(zs[a], gs[b], None, 0.1, label_offset=off) for a, b in pairs])
    (single * w.cuda()).sum().backward()
    close(got, single, 1e-6)
    close(zg.grad, zs.grad, 1e-4)
    close(gg.grad, gs.grad, 1e-4)
    with pytest.raises(ValueError):
        ops.infonce_pairs(zg, 0.1, gathered=gg, label_offset=Ball)      # rows would fall outside the gathered set


def test_random_small_architectures_and_shapes_match_oracle():
    """Seeded random architectures (block counts and widths, dilations, head counts, feature widths) and input shapes
    (batches from 1, windows from 1 timestep, graphs from one isolated service, token-sequence and absent logs) through
    one training forward + backward in both negative modes against the oracle: probabilities 1e-4 relative, identical
    top-k, losses and every gradient 1e-3."""
    import ocean
    from ocean.data import ServiceGraph
    from oracle import ocean_ref as R
    rng = torch.Generator().manual_seed(2718)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    for case in range(32):
        L = ri(1, 4)
        cfg = ocean.OCEANConfig()
        cfg.model.dcnn_channels = [4 * ri(1, 6) for _ in range(L)]
        cfg.model.dcnn_dilation_rates = [pick([1, 2, 3, 4, 8]) for _ in range(L)]
        gh, ah = pick([1, 2, 4]), pick([1, 2, 4])
        hidden = 8 * ri(1, 3) * max(gh, ah)         # the fusion stage takes the graph features at the attention width
        cfg.model.hidden_dim = hidden               # (ocean_model.py:87-101), so the two widths are one
        cfg.model.attention_dim = hidden
        cfg.model.attention_num_heads, cfg.model.gnn_num_heads = ah, gh
        cfg.model.gnn_num_layers = ri(1, 3)
        cfg.model.fusion_projection_dim = 16 * ri(1, 3)
        S = ri(1, 9)
        cfg.model.num_services = S
        F, LD, GF = ri(1, 13), 4 * ri(1, 8), ri(1, 9)
        cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = F, LD, GF
        B, Tn = pick([1, 1, 2, 3, 5, 8]), pick([1, 2, 3, 7, 10, 16])
        negatives = pick(["in_batch", "bank"])
        m = ocean.OCEANModel(cfg)
        zero_dropout(m)
        sd = synth.fill_state_dict(m.state_dict(), 100 + case)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.contrastive_negatives = negatives
        adj = (torch.rand(S, S, generator=rng) < 0.4).float()
        adj.fill_diagonal_(0.0)
        nf = torch.randn(S, GF, generator=rng)
        metrics = torch.randn(B, Tn, F, generator=rng)
        logs_kind = pick(["vector", "vector", "tokens", "none"])
        logs = None if logs_kind == "none" else (torch.randn(B, LD, generator=rng) if logs_kind == "vector" else torch.randn(B, 3, LD, generator=rng))
        labels = torch.zeros(B, S)
        labels[torch.arange(B), torch.randint(0, S, (B,), generator=rng)] = 1.0
        what = (case, cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates, gh, ah, hidden, cfg.model.attention_dim, S, F, LD, GF, B, Tn, negatives, logs_kind)
        rsd = R.require_grad(sd)
        if B * Tn == 1:
            # one value per channel: BatchNorm refuses to train on it, in the reference (torch) and here alike
            with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
                R.model_forward(rsd, metrics, adj, nf, logs, training=True, negatives=negatives, dilations=tuple(cfg.model.dcnn_dilation_rates),
                                gnn_heads=gh, gnn_layers=cfg.model.gnn_num_layers, fusion_heads=8, log_dim=LD)
            with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
                m(metrics, ServiceGraph(adj, nf), logs)
            continue
        ro = R.model_forward(rsd, metrics, adj, nf, logs, training=True, negatives=negatives, dilations=tuple(cfg.model.dcnn_dilation_rates),
                             gnn_heads=gh, gnn_layers=cfg.model.gnn_num_layers, fusion_heads=8, log_dim=LD)
        rl = R.compute_loss(ro, labels)
        rl["total_loss"].backward()
        out = m(metrics, ServiceGraph(adj, nf), logs)
        losses = m.compute_loss(out, labels.cuda())
        losses["total_loss"].backward()
        probs, rprobs = out["root_cause_probs"].detach().cpu(), ro["root_cause_probs"].detach()
        assert ((probs - rprobs).abs() / rprobs).max().item() < 1e-4, what
        k = min(3, S)
        assert torch.equal(torch.topk(probs, k, dim=-1).indices, torch.topk(rprobs, k, dim=-1).indices), what
        for name in ("total_loss", "prediction_loss", "contrastive_loss"):
            try:
                close(losses[name], rl[name], 1e-3)
            except AssertionError as e:
                raise AssertionError(f"{what} {name}: {e}")
        for name, p in m.named_parameters():
            if rsd[name].grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, (what, name)
            else:
                try:
                    close(p.grad, rsd[name].grad, 1e-3)
                except AssertionError as e:
                    raise AssertionError(f"{what} {name}: {e}")
