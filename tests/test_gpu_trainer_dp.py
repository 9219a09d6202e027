"""GPU tests of the Trainer's LossFunction / train_step, the streaming front-end
with the real learner, and 2-rank data parallelism on the HIP path (two
processes sharing the one GPU, gloo transport) against the single-process
full-batch oracle."""
import os
import socket
from datetime import datetime

import numpy as np
import pytest
import torch

from oracle import ocean_ref as R
from oracle import synth
from test_gpu_model import T, build, close, graph_of, tiny_config, zero_dropout

pytestmark = pytest.mark.gpu


def test_trainer_loss_function_matches_reference_fixture(golden):
    from ocean.models.training import LossFunction
    g = golden("model_train_tiny_bank")
    sd = {k[3:]: T(g[k]).clone() for k in g.files if k.startswith("sd/")}
    m = build(tiny_config(), sd).train()
    out = m(T(g["s0/metrics"]), graph_of(g), T(g["s0/logs"]), return_intermediate=True)
    temporal = out["intermediate"]["temporal_features"]
    losses = LossFunction()(out, T(g["s0/labels"]).cuda(), temporal, temporal.detach() * 0.9)
    for k in g.files:
        if k.startswith("trainer_loss/"):
            close(torch.as_tensor(losses[k[13:]]), g[k], 1e-3)
    # gradients of the two extra terms against torch autograd on the same device values
    f = out["intermediate"]["fused_representation"].detach().cpu().requires_grad_(True)
    tm = temporal.detach().cpu().requires_grad_(True)
    ref = R.trainer_loss({"root_cause_probs": out["root_cause_probs"].detach().cpu(), "temporal_features": tm,
                          "fused_representation": f}, T(g["s0/labels"]), tm.detach() * 0.9)
    (0.05 * ref["consistency_loss"] + 0.02 * ref["diversity_loss"]).backward()
    from ocean import ops
    fd = out["intermediate"]["fused_representation"].detach().requires_grad_(True)
    td = temporal.detach().requires_grad_(True)
    (0.05 * ops.mse_loss(td, td.detach() * 0.9) + 0.02 * ops.diversity_loss(fd)).backward()
    close(fd.grad, f.grad, 1e-3)
    close(td.grad, tm.grad, 1e-3)


def test_trainer_train_step_and_epoch_loop(tmp_path):
    import ocean
    from ocean.data import ServiceGraph
    from ocean.models.training import Trainer
    cfg = tiny_config()
    cfg.training.checkpoint_dir = str(tmp_path)
    cfg.training.save_checkpoint_every = 1
    m = ocean.OCEANModel(cfg)
    zero_dropout(m)
    sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7))

    def batch(seed, B=4):
        g = torch.Generator().manual_seed(seed)
        lab = torch.zeros(B, 6)
        lab[torch.arange(B), torch.randint(0, 6, (B,), generator=g)] = 1
        return {"metrics": torch.randn(B, 8, 5, generator=g), "logs": torch.randn(B, 24, generator=g), "labels": lab,
                "graphs": [sg] * B, "timestamps": [datetime(2024, 1, 1)] * B}

    class Loaders:
        def get_all_loaders(self):
            return [batch(1), batch(2), batch(3, B=3)], [batch(4)], [batch(5)]

    tr = Trainer(m, cfg, Loaders())
    first = tr.train_step(batch(0))
    assert set(first) >= {"prediction_loss", "total_contrastive_loss", "consistency_loss", "diversity_loss", "total_loss"}
    assert first["consistency_loss"] == 0.0 and np.isfinite(first["total_loss"])
    second = tr.train_step(batch(0))
    assert second["consistency_loss"] > 0.0
    hist = tr.train(num_epochs=2)
    assert len(hist["train_history"]["train_total_loss"]) == 2 and len(hist["val_history"]["val_total_loss"]) == 2
    assert (tmp_path / "final_model.pt").exists() and (tmp_path / "checkpoint_epoch_2.pt").exists()
    tr.load_checkpoint(str(tmp_path / "checkpoint_epoch_2.pt"))


def test_streaming_handler_drives_real_updates():
    import ocean
    from ocean.data import DatasetSample, ServiceGraph
    from ocean.models.training import OnlineLearner, StreamingHandler
    cfg = tiny_config()
    m = ocean.OCEANModel(cfg).cuda()
    learner = OnlineLearner(m, cfg)
    h = StreamingHandler(learner, max_concurrent_streams=4, processing_interval=0.05)
    sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7))
    for s in range(4):
        assert h.create_stream(f"s{s}")
        for i in range(3):                 # 3 per stream -> one batch of 12 (<= 5 per modality keeps the bank in its safe range or not: ring handles it)
            lab = torch.zeros(6)
            lab[(s + i) % 6] = 1
            h.add_sample_to_stream(f"s{s}", DatasetSample(torch.randn(8, 5), torch.randn(24), sg, lab, datetime.now()))
    w0 = m.prediction_head[0].weight.detach().clone()
    assert h.stream_processor.process_once() == 12
    assert learner.update_count == 1 and not torch.equal(w0, m.prediction_head[0].weight)
    assert h.get_comprehensive_stats()["learner_stats"]["total_updates"] == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, fix, out):
    """One of two ranks sharing the GPU: ocean.parallel.DataParallel itself, with the host-staged gloo transport in
    place of RCCL (two RCCL ranks cannot share a device)."""
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(90, exit=True)       # a stuck rank reports where, instead of hanging the run
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean
        from ocean.data import ServiceGraph
        from ocean.models.training import OnlineLearner, Trainer
        from ocean.parallel import DataParallel, HostStagedTransport
        cfg = tiny_config()
        m = ocean.OCEANModel(cfg)
        zero_dropout(m)
        m.load_state_dict({k: v.clone() for k, v in fix["sd"].items()})
        m = m.cuda()
        m.contrastive_negatives = fix["negatives"]
        if fix["trainer"]:
            learner = Trainer(m, cfg, data_loader=None)
        else:
            learner = OnlineLearner(m, cfg)
        dp = DataParallel(m, learner, transport=HostStagedTransport())
        assert learner.grad_sync is not None and learner.data_parallel is dp and not dp.capturable
        B = fix["metrics"].shape[0] // world
        sl = slice(rank * B, (rank + 1) * B)
        sg = ServiceGraph(fix["adj"], fix["nf"])
        m.train()
        o = m(fix["metrics"][sl], sg, fix["logs"][sl], return_intermediate=True)
        if fix["trainer"]:
            losses = learner.loss_function(o, fix["labels"][sl].cuda(), o["intermediate"]["temporal_features"], None)
        else:
            losses = m.compute_loss(o, fix["labels"][sl].cuda())
        learner.optimizer.zero_grad()
        losses["total_loss"].backward()
        learner.grad_sync(m)
        torch.cuda.synchronize()
        lo, hi = dp.arena.flat.data_ptr(), dp.arena.flat.data_ptr() + 4 * dp.arena.flat.numel()
        cm = m.fusion_module.contrastive_module
        out[rank] = {"probs": o["root_cause_probs"].detach().cpu(), "loss": float(losses["total_loss"].detach()),
                     "grads": {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None},
                     "in_arena": all(lo <= p.grad.data_ptr() < hi for p in m.parameters() if p.grad is not None),
                     "rm": m.temporal_encoder.conv_layers[1].batch_norm.running_mean.detach().cpu(),
                     "bank": cm.memory_bank.detach().cpu(), "ptr": int(cm.memory_ptr),
                     "diversity": float(losses["diversity_loss"]) if fix["trainer"] else 0.0}
    finally:
        dist.destroy_process_group()


def _fixture(B=8):
    import ocean
    cfg = tiny_config()
    sd = synth.fill_state_dict(ocean.OCEANModel(cfg).state_dict(), 21)
    g = torch.Generator().manual_seed(5)
    fix = {"sd": sd, "metrics": torch.randn(B, 12, 5, generator=g), "logs": torch.randn(B, 24, generator=g),
           "adj": synth.random_digraph(6, 2.0, 2), "nf": torch.randn(6, 7, generator=g), "labels": torch.zeros(B, 6),
           "negatives": "in_batch", "trainer": False}
    fix["labels"][torch.arange(B), torch.randint(0, 6, (B,), generator=g)] = 1
    return fix


@pytest.mark.parametrize("negatives,trainer", [("in_batch", False), ("bank", False), ("in_batch", True)])
def test_two_ranks_reproduce_single_process_full_batch(negatives, trainer):
    """Two ranks on half batches, wired by DataParallel: synchronised BatchNorm statistics, gathered InfoNCE positives
    (or, with memory-bank negatives, the gathered ring write), the Trainer's diversity term over the gathered fused
    rows, gradients averaged in the arena - against ONE process on the full batch (the oracle)."""
    import torch.multiprocessing as mp
    fix = _fixture()
    fix["negatives"], fix["trainer"] = negatives, trainer
    rsd = R.require_grad(fix["sd"])
    ro = R.model_forward(rsd, fix["metrics"], fix["adj"], fix["nf"], fix["logs"], training=True, negatives=negatives,
                         dilations=(1, 2), gnn_heads=2, log_dim=24)
    rl = R.trainer_loss(ro, fix["labels"]) if trainer else R.compute_loss(ro, fix["labels"])
    rl["total_loss"].backward()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_dp_worker, args=(2, _free_port(), fix, out), nprocs=2, join=True)
    probs = torch.cat([out[0]["probs"], out[1]["probs"]])
    rel = ((probs - ro["root_cause_probs"]).abs() / ro["root_cause_probs"]).max().item()
    assert rel < 1e-4, rel                                     # synchronised BatchNorm statistics: global-batch forward
    assert abs((out[0]["loss"] + out[1]["loss"]) / 2 - float(rl["total_loss"])) < 1e-3 * abs(float(rl["total_loss"]))
    for k, gref in ((k, v.grad) for k, v in rsd.items() if v.grad is not None):
        for r in (0, 1):
            close(out[r]["grads"][k], gref, 1e-3)              # averaged gradients == full-batch gradients, on both ranks
    assert out[0]["in_arena"] and out[1]["in_arena"]
    close(out[0]["rm"], rsd["temporal_encoder.conv_layers.1.batch_norm.running_mean"], 1e-5)
    if negatives == "bank":
        # both ranks wrote the whole batch's 3 x 8 rows, in the single-process order
        assert out[0]["ptr"] == out[1]["ptr"] == int(rsd["fusion_module.contrastive_module.memory_ptr"])
        assert torch.equal(out[0]["bank"], out[1]["bank"])
        close(out[0]["bank"], rsd["fusion_module.contrastive_module.memory_bank"], 1e-4)
    if trainer:
        for r in (0, 1):
            assert abs(out[r]["diversity"] - float(rl["diversity_loss"])) <= 1e-3 * abs(float(rl["diversity_loss"]))


def _dp_class_worker(rank, world, port, fix, out):
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(90, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import ocean
        from ocean.data import ServiceGraph
        from ocean.models.training import OnlineLearner
        from ocean.parallel import DataParallel
        cfg = tiny_config()
        res = {}
        sg = ServiceGraph(fix["adj"], fix["nf"])
        for mode in ("eager", "graph"):
            m = ocean.OCEANModel(cfg)
            zero_dropout(m)
            m.load_state_dict({k: v.clone() for k, v in fix["sd"].items()})
            m = m.cuda()
            m.contrastive_negatives = "in_batch"
            learner = OnlineLearner(m, cfg)
            dp = DataParallel(m, learner)                  # RCCL transport, early slice on
            assert dp.capturable and dp.early_bucket and 0 < dp._split < dp.arena.flat.numel()
            if mode == "eager":
                m.train()
                o = m(fix["metrics"], sg, fix["logs"])
                losses = m.compute_loss(o, fix["labels"].cuda())
                learner.optimizer.zero_grad()
                losses["total_loss"].backward()
                started = dp._early_work is not None       # the post-encoder slice was launched from inside backward
                before = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
                learner.grad_sync(m)
                torch.cuda.synchronize()
                same = all(torch.equal(before[k], p.grad) for k, p in m.named_parameters() if p.grad is not None)
                res["eager"] = {"started": bool(started), "same": bool(same), "reset": dp._early_work is None,
                                "grads": {k: v.cpu() for k, v in before.items()}, "loss": float(losses["total_loss"].detach())}
            learner.use_hip_graph = mode == "graph"
            hist = [learner.process_tensors(fix["metrics"], sg, fix["logs"], fix["labels"])["batch_loss"] for _ in range(6)]
            res[mode + "_hist"] = hist
            res[mode + "_params"] = [p.detach().cpu() for p in m.parameters()]
            res[mode + "_recorded"] = len(learner._graphed)
            from ocean import _hip
            _hip.unregister_grad_arena(dp.arena)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_data_parallel_class_on_one_rccl_rank_eager_and_recorded():
    """ocean.parallel.DataParallel over RCCL with one rank (all a one-GPU box can host): the gradient arena is reduced
    in two slices, the first started from inside backward; with one rank the average is the identity, so every gradient
    must come back bit-identical - which it would not if the early slice were sent before some gradient in it was
    complete - and equal the oracle's.  Then the whole update step INCLUDING its collectives (BatchNorm statistics,
    gathered positives, the two gradient slices) recorded as one HIP graph and replayed: same losses and parameters as
    the eagerly launched run."""
    import torch.multiprocessing as mp
    fix = _fixture()
    rsd = R.require_grad(fix["sd"])
    ro = R.model_forward(rsd, fix["metrics"], fix["adj"], fix["nf"], fix["logs"], training=True, negatives="in_batch",
                         dilations=(1, 2), gnn_heads=2, log_dim=24)
    rl = R.compute_loss(ro, fix["labels"])
    rl["total_loss"].backward()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_dp_class_worker, args=(1, _free_port(), fix, out), nprocs=1, join=True)
    r = out[0]
    assert r["eager"]["started"] and r["eager"]["same"] and r["eager"]["reset"], r["eager"]
    assert abs(r["eager"]["loss"] - float(rl["total_loss"])) < 1e-3 * abs(float(rl["total_loss"]))
    for k, gref in ((k, v.grad) for k, v in rsd.items() if v.grad is not None):
        close(r["eager"]["grads"][k], gref, 1e-3)
    assert r["eager_recorded"] == 0 and r["graph_recorded"] == 1
    for a, b in zip(r["eager_hist"], r["graph_hist"]):
        assert abs(a - b) <= 1e-4 * abs(a), (r["eager_hist"], r["graph_hist"])
    for a, b in zip(r["eager_params"], r["graph_params"]):
        close(b, a, 3e-3)


def _shard_gpu_worker(rank, world, port, fix, out):
    import faulthandler
    from datetime import datetime as dt
    import torch.distributed as dist
    faulthandler.dump_traceback_later(120, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean
        from ocean.data import DatasetSample, ServiceGraph
        from ocean.models.training import OnlineLearner, StreamingHandler
        from ocean.parallel import DataParallel, HostStagedTransport
        cfg = tiny_config()
        m = ocean.OCEANModel(cfg)
        zero_dropout(m)
        m.load_state_dict({k: v.clone() for k, v in fix["sd"].items()})
        m = m.cuda()
        m.contrastive_negatives = fix["negatives"]
        learner = OnlineLearner(m, cfg)
        dp = DataParallel(m, learner, transport=HostStagedTransport())
        sg = ServiceGraph(fix["adj"], fix["nf"])
        h = StreamingHandler(learner, max_concurrent_streams=4, processing_interval=0.05, service_graph=sg)
        h.stream_processor.batch_size = 4
        for s in range(4):
            h.create_stream(f"s{s}")
        # rank 0 owns s0 (samples 0..3) and s2 (sample 4): chunks of 4 and 1; rank 1 owns s1 (samples 5..7): one chunk of 3
        for i in range(8):
            smp = DatasetSample(fix["metrics"][i], fix["logs"][i], sg, fix["labels"][i], dt(2024, 1, 1))
            h.add_sample_to_stream("s0" if i < 4 else ("s2" if i == 4 else "s1"), smp)
        done = h.stream_processor.process_once()
        torch.cuda.synchronize()
        cm = m.fusion_module.contrastive_module
        out[rank] = {"done": done, "updates": learner.update_count, "hist": list(learner.performance_history["batch_loss"]),
                     "params": {k: v.detach().cpu() for k, v in m.state_dict().items()},
                     "standins": h.stream_processor.standin_steps}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("negatives", ["bank", "in_batch"])
def test_sharded_streams_two_ranks_uneven_arrivals_match_the_union_batches(negatives):
    """Two ranks, uneven arrivals on their streams (5 and 3 samples, chunks of 4): step 0 trains on rank 0's 4 rows and
    rank 1's 3, step 1 on rank 0's last row while rank 1 takes part with a weightless stand-in.  Every coupling is exact
    (BatchNorm statistics over the real rows of both ranks, losses and gradients weighted by row counts; with
    memory-bank negatives the ranks' projections written into the ring in rank order, with in-batch negatives every
    anchor scored against the 7 (then 1) rows of both ranks, gathered by their true counts), so both ranks must end with
    the same state as ONE process that takes two update steps on the union batches [0-3, 5-7] and [4]."""
    import torch.multiprocessing as mp
    fix = _fixture()
    fix["negatives"] = negatives
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_shard_gpu_worker, args=(2, _free_port(), fix, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert (a["done"], b["done"]) == (5, 3) and a["updates"] == b["updates"] == 2
    assert (a["standins"], b["standins"]) == (0, 1)
    assert a["hist"] == b["hist"]                                    # the all-reduced global means, identical on both ranks
    for k in a["params"]:
        assert torch.equal(a["params"][k], b["params"][k]), k        # same averaged gradients, same update, same bank
    # one process on the union batches
    rsd = R.require_grad(fix["sd"])
    tk = R.trainable_keys(rsd)
    state = {k: {} for k in tk}
    ref_hist = []
    for rows in ([0, 1, 2, 3, 5, 6, 7], [4]):
        for k in tk:
            rsd[k].grad = None
        ro = R.model_forward(rsd, fix["metrics"][rows], fix["adj"], fix["nf"], fix["logs"][rows], training=True, negatives=negatives,
                             dilations=(1, 2), gnn_heads=2, log_dim=24)
        rl = R.compute_loss(ro, fix["labels"][rows])
        rl["total_loss"].backward()
        ref_hist.append(float(rl["total_loss"]))
        # the dead query / key projections of the length-1 self-attentions get exact-zero gradients in the reference
        # (and on the HIP path), so Adam's weight decay moves them; the oracle leaves them out of its graph
        for k in tk:
            if "_self_attention.w_q." in k or "_self_attention.w_k." in k:
                rsd[k].grad = torch.zeros_like(rsd[k])
        wg = [k for k in tk if rsd[k].grad is not None]
        R.clip_grad_norm([rsd[k].grad for k in wg], 1.0)
        with torch.no_grad():
            R.adam_step([rsd[k] for k in wg], [rsd[k].grad for k in wg], [state[k] for k in wg])
    for got, ref in zip(a["hist"], ref_hist):
        assert abs(got - ref) <= 1e-3 * abs(ref), (a["hist"], ref_hist)
    c = "fusion_module.contrastive_module."
    assert int(a["params"][c + "memory_ptr"]) == int(rsd[c + "memory_ptr"])
    close(a["params"][c + "memory_bank"], rsd[c + "memory_bank"], 2e-3)
    for k in ("temporal_encoder.conv_layers.1.batch_norm.running_mean", "temporal_encoder.conv_layers.0.batch_norm.running_var"):
        close(a["params"][k], rsd[k], 1e-3)
    for k in tk:
        if rsd[k].is_floating_point():
            # two Adam steps move an element by up to 2 lr = 2e-3 whatever its gradient's size (elements whose true
            # gradient is ~0, like a bias in front of a BatchNorm, can land anywhere in that range); on average the
            # tensors agree far more closely
            d = (a["params"][k].double() - rsd[k].detach().double()).abs()
            assert float(d.max()) <= 2.1e-3 and float(d.mean()) <= 5e-4, (k, float(d.max()), float(d.mean()))


def test_trainer_recorded_step_equals_the_eager_one():
    """Trainer.use_hip_graph: the third train_step with one batch signature is recorded (forward with intermediates,
    LossFunction with its consistency term against the previous batch, backward, clip + Adam) and replayed afterwards.
    Against a second trainer that launches every step eagerly, over a sequence with a ragged batch in the middle (that
    step and the one after it - which has no predecessor of its shape - run eagerly in both): the same loss dictionaries
    step by step and the same parameters at the end."""
    import ocean
    from ocean.data import ServiceGraph
    from ocean.models.training import Trainer
    sg = ServiceGraph(synth.random_digraph(6, 2.0, 1), torch.randn(6, 7, generator=torch.Generator().manual_seed(3)))

    def batch(seed, B=4):
        g = torch.Generator().manual_seed(seed)
        lab = torch.zeros(B, 6)
        lab[torch.arange(B), torch.randint(0, 6, (B,), generator=g)] = 1
        return {"metrics": torch.randn(B, 8, 5, generator=g), "logs": torch.randn(B, 24, generator=g), "labels": lab,
                "graphs": [sg] * B, "timestamps": [datetime(2024, 1, 1)] * B}

    trainers = []
    for use_graph in (True, False):
        cfg = tiny_config()
        m = ocean.OCEANModel(cfg)
        zero_dropout(m)
        m.load_state_dict(synth.fill_state_dict(m.state_dict(), 5))
        m.contrastive_negatives = "in_batch"
        tr = Trainer(m, cfg, data_loader=None)
        tr.use_hip_graph = use_graph
        trainers.append(tr)
    sizes = [4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4]
    for i, B in enumerate(sizes):
        if i == 8:
            for tr in trainers:
                tr.optimizer.param_groups[0]["lr"] = 3e-4          # the epoch scheduler's change must reach the replays
        a, b = (tr.train_step(batch(100 + i, B)) for tr in trainers)
        assert set(a) == set(b)
        for k in b:
            tol = 1e-5 if i < 5 else 2e-3          # Adam amplifies the atomics' rounding noise over the steps
            assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1e-3), (i, k, a[k], b[k])
        if B == 4 and i not in (0, 7):
            assert a["consistency_loss"] > 0.0
    assert len(trainers[0]._graphed) == 1 and not trainers[1]._graphed
    st = trainers[0].optimizer.state
    assert {int(s["step"]) for s in st.values() if s} == {len(sizes)}
    for p, q in zip(trainers[0].model.parameters(), trainers[1].model.parameters()):
        close(p, q, 2e-3)


def _dp_variants_worker(rank, world, port, fix, out):
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(120, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import ocean
        from ocean import _hip
        from ocean.data import ServiceGraph
        from ocean.models.training import OnlineLearner, Trainer
        from ocean.parallel import DataParallel
        cfg = tiny_config()
        sg = ServiceGraph(fix["adj"], fix["nf"])
        B = fix["metrics"].shape[0]
        batch = {"metrics": fix["metrics"], "logs": fix["logs"], "labels": fix["labels"], "graphs": [sg] * B}
        res = {}
        for variant in ("bank_learner", "trainer"):
            for mode in ("eager", "graph"):
                m = ocean.OCEANModel(cfg)
                zero_dropout(m)
                m.load_state_dict({k: v.clone() for k, v in fix["sd"].items()})
                m = m.cuda()
                if variant == "bank_learner":
                    m.contrastive_negatives = "bank"
                    owner = OnlineLearner(m, cfg)
                    step = lambda: owner.process_tensors(fix["metrics"], sg, fix["logs"], fix["labels"])["batch_loss"]
                else:
                    m.contrastive_negatives = "in_batch"
                    owner = Trainer(m, cfg, data_loader=None)
                    step = lambda: owner.train_step(batch)["total_loss"]
                dp = DataParallel(m, owner)
                owner.use_hip_graph = mode == "graph"
                hist = [step() for _ in range(6)]
                cm = m.fusion_module.contrastive_module
                res[(variant, mode)] = {"hist": hist, "params": [p.detach().cpu() for p in m.parameters()],
                                        "recorded": len(owner._graphed), "bank": cm.memory_bank.detach().cpu(), "ptr": int(cm.memory_ptr)}
                _hip.unregister_grad_arena(dp.arena)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_one_rccl_rank_records_the_bank_mode_step_and_the_trainer_step():
    """With an RCCL process group attached (one rank: all a one-GPU box can host) the update step in memory-bank mode (the
    gathered projections written into the ring on the device) and Trainer.train_step (diversity rows gathered) are
    recorded with their collectives inside the HIP graph and give the eager runs' losses, parameters, bank and pointer."""
    import torch.multiprocessing as mp
    fix = _fixture()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_dp_variants_worker, args=(1, _free_port(), fix, out), nprocs=1, join=True)
    r = out[0]
    for variant in ("bank_learner", "trainer"):
        e, g = r[(variant, "eager")], r[(variant, "graph")]
        assert e["recorded"] == 0 and g["recorded"] == 1, variant
        for a, b in zip(e["hist"], g["hist"]):
            assert abs(a - b) <= 1e-4 * abs(a), (variant, e["hist"], g["hist"])
        for a, b in zip(e["params"], g["params"]):
            close(b, a, 3e-3)
        assert e["ptr"] == g["ptr"], variant
        close(g["bank"], e["bank"], 1e-3)
