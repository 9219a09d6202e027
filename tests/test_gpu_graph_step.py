"""The HIP-graph replay of the online update step must be the same computation as the eagerly launched step.

With dropout off the two are compared step by step (same kernels, same order: losses to 1e-5, parameters and Adam
state within the fp32 noise Adam amplifies); with dropout on, replays must draw fresh masks (the recorded seeds are
frozen, the device epoch counter is not), follow the host's learning-rate changes, and keep the optimizer's step
counters in line with the number of updates.
"""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


def _cfg():
    import ocean
    cfg = ocean.OCEANConfig()
    cfg.model.dcnn_channels, cfg.model.dcnn_dilation_rates = [8, 16], [1, 2]
    cfg.model.hidden_dim = cfg.model.attention_dim = 16
    cfg.model.attention_num_heads, cfg.model.gnn_num_heads = 4, 2
    cfg.model.fusion_projection_dim, cfg.model.num_services = 32, 6
    cfg.data.metrics_features, cfg.data.log_embedding_dim, cfg.data.graph_features = 5, 24, 7
    return cfg


def _learner(dropout_off, use_graph, negatives="in_batch"):
    import ocean
    from ocean.models.training import OnlineLearner
    cfg = _cfg()
    m = ocean.OCEANModel(cfg)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 3))
    if dropout_off:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    m = m.cuda()
    m.contrastive_negatives = negatives
    learner = OnlineLearner(m, cfg)
    learner.use_hip_graph = use_graph
    return m, learner


def _data(i, B=4, T=12):
    from ocean.data import ServiceGraph
    g = lambda s: torch.Generator().manual_seed(s)
    metrics, logs = torch.randn(B, T, 5, generator=g(10 + i)).cuda(), torch.randn(B, 24, generator=g(20 + i)).cuda()
    labels = torch.zeros(B, 6)
    labels[torch.arange(B), torch.randint(0, 6, (B,), generator=g(30 + i))] = 1.0
    return metrics, logs, labels.cuda()


def _graph():
    from ocean.data import ServiceGraph
    adj = synth.random_digraph(6, 2.0, 4)
    return ServiceGraph(adjacency_matrix=adj, node_features=torch.randn(6, 7, generator=torch.Generator().manual_seed(1)))


def test_replay_matches_eager_steps():
    sg = _graph()
    runs = {}
    for use_graph in (False, True):
        m, learner = _learner(True, use_graph)
        losses = []
        for i in range(7):
            metrics, logs, labels = _data(i)
            losses.append(learner.process_tensors(metrics, sg, logs, labels))
        assert len(learner._graphed) == (1 if use_graph else 0)
        runs[use_graph] = (losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, learner)
    for a, b in zip(runs[False][0], runs[True][0]):
        assert a["update_count"] == b["update_count"]
        assert abs(a["batch_loss"] - b["batch_loss"]) <= 1e-5 * max(1.0, abs(a["batch_loss"]))
        assert abs(a["batch_prediction_loss"] - b["batch_prediction_loss"]) <= 1e-5
    for k, v in runs[False][1].items():
        w = runs[True][1][k]
        scale = max(v.double().abs().max().item(), 1e-3)
        assert (v.double() - w.double()).abs().max().item() <= 2e-3 * scale, k
    se, sg_ = runs[False][2].optimizer.state_dict()["state"], runs[True][2].optimizer.state_dict()["state"]
    assert se.keys() == sg_.keys()
    for k in se:
        assert int(se[k]["step"]) == int(sg_[k]["step"]) == 7


def test_replays_draw_new_masks_and_follow_the_learning_rate():
    sg = _graph()
    m, learner = _learner(False, True)
    metrics, logs, labels = _data(0)
    for _ in range(3):
        learner.process_tensors(metrics, sg, logs, labels)
    assert len(learner._graphed) == 1
    # learning rate 0 (and no weight decay effect through lr): parameters freeze, yet the loss of identical inputs
    # still changes from replay to replay because every replay draws new dropout masks
    for group in learner.optimizer.param_groups:
        group["lr"] = 0.0
    learner.lr_scheduler.current_lr = 0.0
    learner.lr_scheduler.min_lr = 0.0
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    losses = [learner.process_tensors(metrics, sg, logs, labels)["batch_loss"] for _ in range(4)]
    assert len(set(round(x, 7) for x in losses)) > 1, losses
    for k, v in m.named_parameters():
        assert torch.equal(v, before[k]), k
    # and back on: parameters move again
    for group in learner.optimizer.param_groups:
        group["lr"] = 1e-3
    learner.lr_scheduler.current_lr = 1e-3
    learner.process_tensors(metrics, sg, logs, labels)
    assert any(not torch.equal(v, before[k]) for k, v in m.named_parameters())


def test_ineligible_steps_run_eagerly_and_say_so_once(caplog):
    import logging
    sg = _graph()
    m, learner = _learner(True, True, negatives="bank")
    metrics, logs, labels = _data(0, B=2)
    for _ in range(4):
        learner.process_tensors(metrics, sg, logs, labels)
    assert len(learner._graphed) == 1          # memory-bank mode is recorded too: the ring pointer lives on the device
    # an optimizer whose step cannot be captured: the update runs as separate launches, with ONE warning that says why
    m, learner = _learner(True, True)
    learner.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
    with caplog.at_level(logging.WARNING, logger="ocean.models.training.online_learner"):
        for _ in range(4):
            learner.process_tensors(metrics, sg, logs, labels)
    said = [r for r in caplog.records if "separate launches" in r.getMessage()]
    assert len(learner._graphed) == 0 and len(said) == 1 and "FusedAdam" in said[0].getMessage()
    m, learner = _learner(True, True)
    for i in range(4):
        learner.process_tensors(*(lambda d: (d[0], sg, d[1], d[2]))(_data(i, B=3 + i)))   # a new batch size every step
    assert len(learner._graphed) == 0 and learner.update_count == 4


def test_recorded_steps_are_bounded_least_recently_used_first():
    sg = _graph()
    m, learner = _learner(True, True)
    learner.max_graphs = 2
    data = {B: _data(B, B=B) for B in (2, 3, 4)}
    for B in (2, 3, 4):
        for _ in range(4):                       # two eager steps, then recorded, then replayed
            learner.process_tensors(data[B][0], sg, data[B][1], data[B][2])
        assert len(learner._graphed) == min(2, B - 1)
    shapes = [k[0][0] for k in learner._graphed]
    assert shapes == [3, 4]                      # the batch-2 recording was dropped first
    learner.process_tensors(data[3][0], sg, data[3][1], data[3][2])      # a hit moves batch 3 to the back
    assert [k[0][0] for k in learner._graphed] == [4, 3]
    r = learner.process_tensors(data[2][0], sg, data[2][1], data[2][2])  # seen before: recorded again at once, batch 4 goes
    assert [k[0][0] for k in learner._graphed] == [3, 2] and r["update_count"] == 14


def test_recordings_do_not_survive_a_checkpoint_load(tmp_path):
    """load_checkpoint replaces the optimizer's moment tensors and step counts (online_learner.py:543-565 does the
    same through torch.optim.Adam.load_state_dict): a step recorded before the load would keep updating the old
    tensors.  Save after 4 steps, go on to 6, load, go on to 8: the replaying learner must land where the eager one does."""
    sg = _graph()
    final = {}
    for use_graph in (False, True):
        m, learner = _learner(True, use_graph)
        path = str(tmp_path / f"ckpt_{use_graph}.pt")
        for i in range(6):
            learner.process_tensors(*(lambda d: (d[0], sg, d[1], d[2]))(_data(i)))
            if i == 3:
                learner.save_checkpoint(path)
        n_before = len(learner._graphed)
        learner.load_checkpoint(path)
        assert learner.update_count == 4
        for i in range(4, 8):
            r = learner.process_tensors(*(lambda d: (d[0], sg, d[1], d[2]))(_data(i)))
        assert r["update_count"] == 8
        if use_graph:
            assert n_before == 1 and len(learner._graphed) == 1
            assert next(iter(learner._graphed.values())).generation == learner.optimizer.generation == 1
        st = learner.optimizer.state_dict()["state"]
        assert all(int(v["step"]) == 8 for v in st.values())
        final[use_graph] = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in final[False].items():
        w = final[True][k]
        scale = max(v.double().abs().max().item(), 1e-3)
        assert (v.double() - w.double()).abs().max().item() <= 2e-3 * scale, k


def test_freezing_a_component_is_not_replayed_over():
    """OCEANModel.freeze_component (ocean_model.py:410-452) after a step has been recorded: the recording updates every
    parameter it was recorded with, so the frozen set is part of what a recording is keyed on."""
    sg = _graph()
    m, learner = _learner(True, True)
    metrics, logs, labels = _data(0)
    for _ in range(4):
        learner.process_tensors(metrics, sg, logs, labels)
    assert len(learner._graphed) == 1
    m.freeze_component("temporal")
    frozen = {k: v.detach().clone() for k, v in m.temporal_encoder.named_parameters()}
    head = m.prediction_head[0].weight.detach().clone()
    for _ in range(4):
        learner.process_tensors(metrics, sg, logs, labels)
    assert len(learner._graphed) == 2                      # a second recording, for the new trainable set
    for k, v in m.temporal_encoder.named_parameters():
        assert torch.equal(v, frozen[k]), k
    assert not torch.equal(m.prediction_head[0].weight, head)


def test_replayed_eval_forward_equals_the_eager_one_and_sees_updated_weights():
    """OCEANModel.use_hip_graph: predict_root_cause / an eval-mode forward under no_grad is recorded on its third call
    with one signature and replayed afterwards.  Replays on new inputs (device and host tensors) give the eager
    outputs, a training step in between is seen (the recording reads the parameters in place), another batch size gets
    its own recording, and a forward that needs gradients or intermediates is never replayed."""
    m, learner = _learner(True, False)
    sg = _graph()
    ref_model, _ = _learner(True, False)
    ref_model.load_state_dict(m.state_dict())
    m.use_hip_graph = True
    m.eval(); ref_model.eval()

    def check(i, B=4, host=False):
        metrics, logs, _ = _data(i, B=B)
        if host:
            metrics, logs = metrics.cpu(), logs.cpu()
        got = m.predict_root_cause(metrics, sg, logs, top_k=3)
        ref = ref_model.predict_root_cause(metrics, sg, logs, top_k=3)
        assert torch.equal(got["probabilities"], ref["probabilities"]), i          # same kernels, same order: bit-equal
        assert torch.equal(got["top_k_indices"], ref["top_k_indices"])
        return got

    for i in range(6):
        check(i, host=(i == 4))
    assert len(m._graphed_fwd) == 1
    held = check(6)["probabilities"]
    snapshot = held.clone()
    check(7)
    assert torch.equal(held, snapshot)                    # a later replay does not overwrite what an earlier call returned
    # a training step on both models, then predictions again
    for mod, lr in ((m, learner), (ref_model, None)):
        mod.train()
    from ocean.models.training import OnlineLearner
    ref_learner = OnlineLearner(ref_model, learner.config)
    metrics, logs, labels = _data(50)
    learner.process_tensors(metrics, sg, logs, labels)
    ref_learner.process_tensors(metrics, sg, logs, labels)
    m.eval(); ref_model.eval()
    with torch.no_grad():
        a = m(metrics, sg, logs)["root_cause_probs"]
        b = ref_model(metrics, sg, logs)["root_cause_probs"]
    assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item()      # the two training steps differ by atomics' order
    assert len(m._graphed_fwd) == 1
    for i in range(8, 12):
        check(i, B=2)
    assert len(m._graphed_fwd) == 2
    # gradients wanted, or intermediates: ordinary launches
    out = m(metrics, sg, logs, return_intermediate=True)
    assert "intermediate" in out
    out = m(metrics, sg, logs)
    assert out["root_cause_probs"].requires_grad
    m.train()
    with torch.no_grad():
        m(metrics, sg, logs)
    assert len(m._graphed_fwd) == 2
