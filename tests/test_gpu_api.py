"""Secondary API surface of the reference's classes on the HIP path: per-layer accessors, reduction='none',
checkpoint round trips under torch's restricted unpickler, the recorded update step's keying on the graph's
tensors, and the optimizer's global clipping norm over several parameter groups."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ocean_ref as R
from oracle import synth

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e}"


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if m.__class__.__name__ == "GATConv":
            m.dropout = 0.0


@pytest.mark.parametrize("training", [False, True])
def test_get_temporal_features_per_block(training):
    """DilatedCNN.get_temporal_features (dilated_cnn.py:243-277): time-pooled output of every dilated block."""
    from ocean.models.components import DilatedCNN
    m = DilatedCNN(12, [16, 32, 64], 3, [1, 2, 4], dropout=0.0, output_dim=24)
    sd = synth.fill_state_dict(m.state_dict(), 3)
    m.load_state_dict(sd)
    m = m.cuda().train(training)
    x = torch.randn(5, 20, 12, generator=torch.Generator().manual_seed(1))
    col = {}
    R.dcnn_forward({"t." + k: v.clone() for k, v in sd.items()}, "t.", x, [1, 2, 4], training, 3, collect=col)
    feats = m.get_temporal_features(x.cuda())
    assert len(feats) == 3 and [f.shape[1] for f in feats] == [16, 32, 64]
    for i, f in enumerate(feats):
        close(f, col[f"block{i}"].mean(dim=2), 1e-4, f"block {i}")


def test_get_node_embeddings():
    """GraphNeuralNetwork.get_node_embeddings (graph_neural_network.py:320-356): last layer's node rows, no pooling."""
    from ocean.models.components import GraphNeuralNetwork
    N = 23
    m = GraphNeuralNetwork(12, 64, 2, 4, dropout=0.0, batch_norm=False)
    sd = synth.fill_state_dict(m.state_dict(), 4)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    ei = R.edge_index_of(synth.random_digraph(N, 3.0, 5, self_loop_at=1))
    nf = torch.randn(N, 12, generator=torch.Generator().manual_seed(1))
    col = {}
    R.gnn_forward(sd, "", nf, ei, 1, 4, 2, collect=col)
    emb = m.get_node_embeddings(nf.cuda(), ei.cuda())
    assert emb.shape == (N, 16)
    close(emb, col["node_embeddings"], 1e-4)


@pytest.mark.parametrize("bank", [False, True])
def test_infonce_reduction_none(bank):
    """InfoNCELoss(reduction='none') (graph_fusion.py:100,160): one value per anchor row, and its gradient under a
    non-uniform weighting of the rows."""
    from ocean.models.components import InfoNCELoss
    B, P = 37, 24
    g = torch.Generator().manual_seed(2)
    a, p, n = torch.randn(B, P, generator=g), torch.randn(B, P, generator=g), (torch.randn(16, P, generator=g) if bank else None)
    w = torch.randn(B, generator=g)
    ar, pr = a.clone().requires_grad_(True), p.clone().requires_grad_(True)
    ref = R.infonce(ar, pr, n, 0.1, reduction="none")
    (ref * w).sum().backward()
    ag, pg = a.clone().cuda().requires_grad_(True), p.clone().cuda().requires_grad_(True)
    got = InfoNCELoss(0.1, reduction="none")(ag, pg, None if n is None else n.cuda())
    assert got.shape == (B,)
    (got * w.cuda()).sum().backward()
    close(got, ref, 1e-5)
    close(ag.grad, ar.grad, 1e-3)
    close(pg.grad, pr.grad, 1e-3)
    close(InfoNCELoss(0.1, reduction="sum")(ag, pg, None if n is None else n.cuda()), ref.sum(), 1e-5)
    with pytest.raises(ValueError):
        InfoNCELoss(0.1, reduction="median")(ag, pg)


def test_compute_loss_reduction_none():
    """OCEANModel.compute_loss(reduction='none') (ocean_model.py:249-253): per-element BCE [B, S]; the scalar contrastive
    term broadcasts over it in total_loss, as in the reference."""
    import ocean
    g = torch.Generator().manual_seed(3)
    cfg = ocean.default_config()
    cfg.model.num_services = 7
    m = ocean.OCEANModel(cfg)
    probs = torch.rand(5, 7, generator=g).clamp(1e-4, 1 - 1e-4)
    probs[0, 0], probs[0, 1] = 0.0, 1.0                      # the clamped log terms
    y = (torch.rand(5, 7, generator=g) < 0.3).float()
    w = torch.randn(5, 7, generator=g)
    pr = probs.clone().requires_grad_(True)
    ref = F.binary_cross_entropy(pr, y, reduction="none")
    (ref * w).sum().backward()
    pg = probs.clone().cuda().requires_grad_(True)
    c = torch.tensor(0.25, device="cuda")
    losses = m.compute_loss({"root_cause_probs": pg, "contrastive_losses": {"a_b_contrastive": c}}, y.cuda(), reduction="none")
    assert losses["prediction_loss"].shape == (5, 7) and losses["total_loss"].shape == (5, 7)
    close(losses["prediction_loss"], ref, 1e-6)
    close(losses["total_loss"], ref + 0.1 * 0.25, 1e-6)
    (losses["prediction_loss"] * w.cuda()).sum().backward()
    finite = torch.isfinite(pr.grad)
    close(pg.grad.cpu()[finite], pr.grad[finite], 1e-4)


def _tiny_learner(seed=7, dropout_off=True):
    import ocean
    from ocean.models.training import OnlineLearner
    cfg = ocean.OCEANConfig()
    cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
    cfg.model.hidden_dim = cfg.model.attention_dim = 16
    cfg.model.attention_num_heads, cfg.model.gnn_num_heads = 4, 2
    cfg.model.fusion_projection_dim, cfg.model.num_services = 32, 6
    cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = 5, 24, 7
    m = ocean.OCEANModel(cfg)
    if dropout_off:
        zero_dropout(m)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), seed))
    m = m.cuda()
    m.contrastive_negatives = "in_batch"
    return cfg, m, OnlineLearner(m, cfg)


def _tiny_batch(seed, B=4):
    g = torch.Generator().manual_seed(seed)
    labels = torch.zeros(B, 6)
    labels[torch.arange(B), torch.randint(0, 6, (B,), generator=g)] = 1.0
    return torch.randn(B, 12, 5, generator=g).cuda(), torch.randn(B, 24, generator=g).cuda(), labels.cuda()


def test_checkpoints_load_under_the_restricted_unpickler(tmp_path):
    """save_checkpoint / load_checkpoint of the learner and the trainer with torch.load(weights_only=True)."""
    from ocean.data import ServiceGraph
    from ocean.models.training import Trainer
    cfg, m, learner = _tiny_learner()
    sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7))
    for s in range(3):
        learner.process_tensors(*_tiny_batch(s)[:1], sg, *_tiny_batch(s)[1:])
    path = str(tmp_path / "learner.pt")
    learner.save_checkpoint(path)
    torch.load(path, weights_only=True)                      # raises on anything but tensors and plain containers
    cfg2, m2, l2 = _tiny_learner(seed=8)
    l2.load_checkpoint(path)
    assert l2.update_count == 3 and l2.performance_history["batch_loss"] == learner.performance_history["batch_loss"]
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # the restored optimizer continues exactly like the original
    b3 = _tiny_batch(3)
    r1 = learner.process_tensors(b3[0], sg, b3[1], b3[2])
    r2 = l2.process_tensors(b3[0], sg, b3[1], b3[2])
    assert abs(r1["batch_loss"] - r2["batch_loss"]) < 1e-6
    for a, b in zip(m.parameters(), m2.parameters()):
        close(a, b, 1e-5)           # two runs of one step differ by the atomics' order (~1e-6); a state that was not restored differs by ~lr = 1e-3
    # Trainer
    cfg.training.checkpoint_dir = str(tmp_path / "ck")
    tr = Trainer(m, cfg, data_loader=None)
    tr.train_step({"metrics": b3[0], "logs": b3[1], "labels": b3[2], "graphs": [sg]})
    tr.train_history["train_total_loss"].append(1.5)
    tr._save_checkpoint(0, 0.5)
    ck = str(tmp_path / "ck" / "checkpoint_epoch_1.pt")
    torch.load(ck, weights_only=True)
    tr2 = Trainer(m2, cfg, data_loader=None)
    tr2.load_checkpoint(ck)
    assert tr2.train_history["train_total_loss"] == [1.5]


def test_recorded_step_follows_a_replaced_adjacency_tensor():
    """A recorded (HIP-graph) update step bakes the device CSR in.  Replacing service_graph.adjacency_matrix by a new
    tensor (same graph object, version counter back at 0) must not replay the old topology: the result has to equal
    the eager step on the new graph."""
    from ocean.data import ServiceGraph
    results = {}
    for mode in ("graph", "eager"):
        cfg, m, learner = _tiny_learner(seed=11)
        learner.use_hip_graph = mode == "graph"
        sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7, generator=torch.Generator().manual_seed(0)))
        for s in range(5):
            b = _tiny_batch(s)
            learner.process_tensors(b[0], sg, b[1], b[2])
        if mode == "graph":
            assert len(learner._graphed) == 1
        sg.adjacency_matrix = synth.random_digraph(6, 3.0, 2).clone()   # a NEW tensor
        assert sg.adjacency_matrix._version == 0
        for s in range(5, 9):
            b = _tiny_batch(s)
            r = learner.process_tensors(b[0], sg, b[1], b[2])
        results[mode] = (r["batch_loss"], [p.detach().clone() for p in m.parameters()])
    # nine Adam steps amplify the atomics' rounding noise on near-zero gradients; a replay of the OLD topology is off by
    # orders of magnitude more (checked: ~1e-1 on the loss)
    assert abs(results["graph"][0] - results["eager"][0]) <= 1e-3 * abs(results["eager"][0])
    for a, b in zip(results["graph"][1], results["eager"][1]):
        close(a, b, 2e-3)


def test_clip_norm_is_global_over_parameter_groups():
    """clip_grad_norm_(model.parameters()) takes ONE norm over all gradients; FusedAdam with two parameter groups
    (different learning rates) must clip both by it and then update like torch.optim.Adam."""
    from ocean.optim import FusedAdam
    g = torch.Generator().manual_seed(4)
    ps = [torch.randn(300, generator=g), torch.randn(17, 9, generator=g), torch.randn(5000, generator=g)]
    gs = [torch.randn(p.shape, generator=g) for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    got = [p.clone().cuda().requires_grad_(True) for p in ps]
    ropt = torch.optim.Adam([{"params": ref[:2], "lr": 1e-2}, {"params": ref[2:], "lr": 3e-3}], weight_decay=1e-4)
    fopt = FusedAdam([{"params": got[:2], "lr": 1e-2}, {"params": got[2:], "lr": 3e-3}], weight_decay=1e-4)
    for step in range(3):
        for r, q, gr in zip(ref, got, gs):
            r.grad = gr * (step + 1)
            q.grad = (gr * (step + 1)).cuda()
        norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        fopt.step(max_norm=1.0)
        assert abs(float(fopt.grad_norm()) - float(norm)) <= 1e-5 * float(norm)
        for r, q in zip(ref, got):
            close(q, r, 1e-5)
            close(q.grad, r.grad, 1e-5)


def test_self_resetting_adam_step_matches_torch_adam():
    """FusedAdam with the learning rate and the step count on the device (what a recorded step uses): the norm
    accumulator is cleared and the count advanced by the update kernel itself (ocean_clip_adam_step_auto); five steps
    with a changing learning rate against clip_grad_norm_ + torch.optim.Adam, with and without clipping."""
    from ocean.optim import FusedAdam
    for max_norm in (1.0, 0.0):
        g = torch.Generator().manual_seed(9)
        ps = [torch.randn(300, generator=g), torch.randn(17, 9, generator=g), torch.randn(70000, generator=g)]
        gs = [torch.randn(p.shape, generator=g) for p in ps]
        ref = [p.clone().requires_grad_(True) for p in ps]
        got = [p.clone().cuda().requires_grad_(True) for p in ps]
        ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-4)
        fopt = FusedAdam(got, lr=1e-2, weight_decay=1e-4)
        for step in range(5):
            for r, q, gr in zip(ref, got, gs):
                r.grad = gr * (step + 1)
                q.grad = (gr * (step + 1)).cuda()
            if step == 1:
                fopt.make_capturable(torch.device("cuda"))        # from here on: device-side lr / step, self-resetting kernels
            if step == 3:
                for o in (ropt, fopt):
                    o.param_groups[0]["lr"] = 3e-3
            norm = torch.nn.utils.clip_grad_norm_(ref, max_norm) if max_norm > 0 else None
            ropt.step()
            fopt.step(max_norm=max_norm)
            if max_norm > 0:
                assert abs(float(fopt.grad_norm()) - float(norm)) <= 1e-5 * float(norm)
            for r, q in zip(ref, got):
                close(q, r, 1e-5)
        assert fopt._step_dev is not None and int(fopt._step_dev.item()) == 5
        assert float(fopt._sq[0]) == 0.0 and int(fopt._sq[2:3].view(torch.int32)[0]) == 0     # left clean for the next step


@pytest.mark.parametrize("n,n_c,reduction", [(7, 3, "mean"), (128 * 256, 3, "mean"), (1024 * 256 + 5, 3, "mean"), (300000, 0, "sum"), (4096, 1, "sum")])
def test_prediction_and_total_matches_separate_calls(n, n_c, reduction):
    """ocean_pred_total_fwd/_bwd (one launch each way) against binary_cross_entropy + combine_losses and against
    torch's F.binary_cross_entropy, values and gradients, twice in a row (the finished-block counter must be back at
    zero), including more elements than 512 blocks of 256 threads hold at one each."""
    import torch.nn.functional as F
    from ocean import ops
    g = torch.Generator().manual_seed(n)
    for rep in range(2):
        logits = torch.randn(n, generator=g) * 3
        y = (torch.rand(n, generator=g) < 0.1).float()
        c = torch.rand(n_c, generator=g) if n_c else None
        p_ref = torch.sigmoid(logits).requires_grad_(True)
        c_ref = None if c is None else c.clone().requires_grad_(True)
        pred_ref = F.binary_cross_entropy(p_ref, y, reduction=reduction)
        csum_ref = 0.0
        for q in range(n_c):
            csum_ref = csum_ref + c_ref[q]
        total_ref = 0.7 * pred_ref + 0.1 * csum_ref
        (total_ref + 0.5 * pred_ref).backward()
        p = torch.sigmoid(logits).cuda().requires_grad_(True)
        cg = None if c is None else c.cuda().requires_grad_(True)
        total, pred, csum, vec = ops.prediction_and_total(p.view(-1, 1), y.cuda().view(-1, 1), reduction, cg, 0.7, 0.1)
        (total + 0.5 * pred).backward()
        assert abs(float(pred) - float(pred_ref)) <= 2e-5 * abs(float(pred_ref))
        assert abs(float(total) - float(total_ref)) <= 2e-5 * abs(float(total_ref))
        assert vec.tolist() == [float(total), float(pred), float(csum)]
        close(p.grad, p_ref.grad, 1e-5)
        if n_c:
            close(cg.grad, c_ref.grad, 1e-6)
            # same block-ordered arithmetic on every run
            total2 = ops.prediction_and_total(p.detach().view(-1, 1), y.cuda().view(-1, 1), reduction, cg.detach(), 0.7, 0.1)[0]
            assert float(total2) == float(total)


def test_replayed_steps_leave_a_frozen_components_step_count_alone():
    """After freeze_component, torch.optim.Adam no longer advances 'step' of the frozen parameters (they get no
    gradient); a recorded step must keep the same bookkeeping."""
    from ocean.data import ServiceGraph
    cfg, m, learner = _tiny_learner(seed=12)
    learner.use_hip_graph = True
    sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7, generator=torch.Generator().manual_seed(0)))
    for s in range(2):
        b = _tiny_batch(s)
        learner.process_tensors(b[0], sg, b[1], b[2])
    m.freeze_component("prediction")
    for s in range(2, 8):
        b = _tiny_batch(s)
        learner.process_tensors(b[0], sg, b[1], b[2])
    assert len(learner._graphed) == 1
    st = learner.optimizer.state
    frozen = m.prediction_head[0].weight
    live = m.fusion_module.output_projection[0].weight
    assert int(st[frozen]["step"]) == 2 and int(st[live]["step"]) == 8


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 6, 8, 10])
def test_memory_bank_ring_matches_reference_then_keeps_going(golden, B):
    """ContrastiveLearningModule._update_memory_bank with the pointer on the device (ocean_bank_ring_write) against the
    bank contents and pointer values recorded from the reference (graph_fusion.py:251-273) for every step it survives,
    including the out-of-range pointer value 16; beyond that the ring simply continues."""
    import numpy as np
    from ocean.models.components import ContrastiveLearningModule
    T = lambda a: torch.from_numpy(np.asarray(a))
    g = golden(f"bank_trace_B{B}")
    cm = ContrastiveLearningModule({"temporal": 8, "spatial": 8, "log": 8}, 4, 0.1).cuda()
    cm.memory_bank.copy_(T(g["bank0"]))
    for st in range(len(g["ptrs"])):
        rows = T(g["fed"][st]).cuda()
        cm._update_memory_bank({"temporal": rows[:B], "spatial": rows[B:2 * B], "log": rows[2 * B:]})
        assert int(cm.memory_ptr) == int(g["ptrs"][st])
        assert torch.equal(cm.memory_bank.cpu(), T(g["banks"][st]))
    for _ in range(6):            # where the reference raises (ptr + 3B > 32) the ring simply continues
        cm._update_memory_bank({k: torch.randn(B, 4, device="cuda") for k in ("temporal", "spatial", "log")})
    assert 0 <= int(cm.memory_ptr) <= 16
    big = torch.randn(3, 11, 4, device="cuda")                                    # 33 rows > 16: the last 16 survive
    cm._update_memory_bank({"temporal": big[0], "spatial": big[1], "log": big[2]}, stacked=big)
    assert sorted(map(tuple, cm.memory_bank.tolist())) == sorted(map(tuple, big.reshape(33, 4)[-16:].tolist()))
    sd = cm.state_dict()
    sd["memory_ptr"] = torch.tensor([5])
    cm.load_state_dict(sd)
    cm._update_memory_bank({k: torch.randn(1, 4, device="cuda") for k in ("temporal", "spatial", "log")})
    assert int(cm.memory_ptr) == 8


@pytest.mark.parametrize("B", [1, 4, 7])
def test_memory_bank_mode_replays_as_a_hip_graph(B):
    """The reference's default contrastive mode (memory-bank negatives) and its per-sample update (B = 1,
    online_learner.py:292-360) under use_hip_graph: the ring pointer advances on the device inside the recorded step, so
    bank, pointer, losses and parameters follow the eagerly launched run."""
    from datetime import datetime
    from ocean.data import DatasetSample, ServiceGraph
    runs = {}
    for use_graph in (False, True):
        cfg, m, learner = _tiny_learner(seed=13)
        m.contrastive_negatives = "bank"
        learner.use_hip_graph = use_graph
        sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7, generator=torch.Generator().manual_seed(0)))
        cm = m.fusion_module.contrastive_module
        trace = []
        for s in range(7):
            b = _tiny_batch(s, B)
            if B == 1:
                r = learner.process_sample(DatasetSample(b[0][0].cpu(), b[1][0].cpu(), sg, b[2][0].cpu(), datetime(2024, 1, 1)))
                loss = r["loss"]
            else:
                loss = learner.process_tensors(b[0], sg, b[1], b[2])["batch_loss"]
            trace.append((loss, int(cm.memory_ptr), cm.memory_bank.detach().clone()))
        assert len(learner._graphed) == (1 if use_graph else 0)
        assert learner.update_count == 7
        runs[use_graph] = (trace, [p.detach().clone() for p in m.parameters()])
    for (la, pa, ba), (lb, pb, bb) in zip(runs[False][0], runs[True][0]):
        assert pa == pb
        assert abs(la - lb) <= 1e-4 * abs(la)
        close(bb, ba, 1e-3)
    for a, b in zip(runs[False][1], runs[True][1]):
        close(b, a, 3e-3)


@pytest.mark.parametrize("B", [1, 2, 3, 37, 64, 1000, 1024, 4096])
def test_device_permutations(B):
    """ocean_draw_permutations (torch.randperm's role in the mutual-information estimate, graph_fusion.py:494): every row
    is a permutation of 0..B-1, rows are independent draws, the draw follows the seed and the device epoch counter."""
    import ocean
    from ocean import _hip
    from ocean.models.components import GraphFusionModule
    dev = torch.device("cuda")
    torch.manual_seed(5)
    a = GraphFusionModule.draw_permutations(B, dev)
    assert a.shape == (3, B) and a.dtype == torch.long
    want = torch.arange(B, device=dev)
    for m in range(3):
        assert torch.equal(a[m].sort().values, want)
    torch.manual_seed(5)
    assert torch.equal(GraphFusionModule.draw_permutations(B, dev), a)          # same seed, same draw
    if B >= 37:
        assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])
        assert not torch.equal(GraphFusionModule.draw_permutations(B, dev), a)  # the generator moved on
        torch.manual_seed(5)
        _hip.seed_epoch(dev).add_(1)
        assert not torch.equal(GraphFusionModule.draw_permutations(B, dev), a)  # a replay's epoch bump redraws
        _hip.seed_epoch(dev).sub_(1)
    if B == 4096:
        # roughly uniform: element 0 lands in each quarter of the positions about equally often over many draws
        pos = torch.stack([(GraphFusionModule.draw_permutations(B, dev) == 0).nonzero()[:, 1] for _ in range(200)]).flatten()
        hist = torch.histc(pos.float(), bins=4, min=0, max=B)
        assert hist.min() > 100 and hist.max() < 200, hist
        out = torch.empty(3, B + 1, device=dev, dtype=torch.long)
        st = _hip.lib().ocean_draw_permutations(3, B + 1, 1, None, out.data_ptr(), _hip.stream())
        assert st != 0                                                          # larger draws are the caller's


def test_fused_loss_arithmetic_matches_the_separate_operations():
    """compute_loss's one-launch total (ocean_loss_combine_fwd / _bwd) against the reference's chain of scalar
    operations (ocean_model.py:257-266): bit-equal values, equal gradients, and the read-back vector."""
    from ocean import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    for wp, wc in [(1.0, 0.1), (0.7, 0.3), (1.0, 0.0)]:
        pred = torch.rand((), generator=g).to(dev).requires_grad_()
        c = torch.rand(3, generator=g).to(dev).requires_grad_()
        total, csum, vec = ops.combine_losses(pred, c, wp, wc)
        pr, cr = pred.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
        tc = 0.0
        for q in range(3):
            tc = tc + cr[q]
        tr = wp * pr + wc * tc
        assert torch.equal(total.detach(), tr.detach()) and torch.equal(csum.detach(), tc.detach())
        assert torch.equal(vec, torch.stack([tr.detach(), pr.detach(), tc.detach()]))
        (2.0 * total + 0.5 * csum).backward()
        (2.0 * tr + 0.5 * tc).backward()
        assert torch.allclose(pred.grad, pr.grad, rtol=1e-6, atol=0) and torch.allclose(c.grad, cr.grad, rtol=1e-6, atol=0)
    # no contrastive terms (evaluation): total = wp * pred
    pred = torch.tensor(0.37, device=dev, requires_grad=True)
    total, csum, vec = ops.combine_losses(pred, None, 0.9, 0.1)
    assert torch.equal(total.detach(), (0.9 * pred).detach()) and float(csum.detach()) == 0.0
    total.backward()
    assert abs(float(pred.grad) - 0.9) < 1e-7
