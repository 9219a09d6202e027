"""Generate tests/golden/*.npz by running the reference's own classes.

Runs ONLY in the build container (needs /root/reference); the fixtures it
writes are committed, the reference never travels.  Usage:

    python -m oracle.gen_golden            # from the repo root

Each fixture stores inputs, (small) parameter sets or the seed that
regenerates them through oracle.synth.fill_state_dict, and the reference's
outputs / gradients / post-step state.  `meta_*` entries record which harness
patches were active (dropout zeroed; bank update / negatives patches).
"""
from __future__ import annotations

import importlib.util
import logging
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OCEAN_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from oracle import pyg_shim, synth  # noqa: E402


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if m.__class__.__name__ == "GATConv":
            m.dropout = 0.0


def npd(d, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in d.items()}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrs)} arrays)")


def rnd(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------ DCNN
def gen_dcnn():
    dc = load_by_path("ref_dilated_cnn", "ocean/models/components/dilated_cnn.py")
    for tag, (cin, ch, dil, od, B, T) in {
        "tiny": (5, [8, 16], [1, 3], 12, 3, 10),
        "three": (12, [16, 16, 32], [1, 2, 4], None, 2, 7),
        "t2": (4, [8, 8], [1, 2], 8, 2, 2),
    }.items():
        m = dc.DilatedCNN(cin, ch, 3, dil, dropout=0.0, output_dim=od)
        sd0 = synth.fill_state_dict(m.state_dict(), 100)
        m.load_state_dict(sd0)
        x = rnd((B, T, cin), 1)
        x.requires_grad_(True)
        gw = rnd((B, od or ch[-1]), 2)
        m.train()
        y = m(x)
        (y * gw).sum().backward()
        arrs = npd(sd0, "sd/")
        arrs.update(x=x.detach().numpy(), gw=gw.numpy(), y_train=y.detach().numpy(), dx=x.grad.numpy(),
                    channels=np.array(ch), dilations=np.array(dil))
        arrs.update({"grad/" + k: p.grad.numpy() for k, p in m.named_parameters()})
        arrs.update(npd(m.state_dict(), "sd_after/"))
        m.eval()
        with torch.no_grad():
            arrs["y_eval"] = m(x).numpy()
        save("dcnn_" + tag, **arrs)
    # default architecture: parameters regenerated from the seed, digests of grads
    m = dc.DilatedCNN(12, None, 3, None, dropout=0.0, output_dim=128)
    sd0 = synth.fill_state_dict(m.state_dict(), 101)
    m.load_state_dict(sd0)
    x = rnd((2, 24, 12), 3)
    gw = rnd((2, 128), 4)
    m.train()
    y = m(x)
    (y * gw).sum().backward()
    arrs = dict(x=x.numpy(), gw=gw.numpy(), y_train=y.detach().numpy(), meta_seed=np.array(101))
    arrs.update({"gdig/" + k: synth.digest(p.grad).numpy() for k, p in m.named_parameters()})
    arrs.update({"sd_after/" + k: v.numpy() for k, v in m.state_dict().items() if "running" in k})
    m.eval()
    with torch.no_grad():
        arrs["y_eval"] = m(x).numpy()
    save("dcnn_default", **arrs)
    # the reference's training-mode failure for one value per channel
    try:
        m.train()
        m(rnd((1, 1, 12), 5))
        raised = ""
    except ValueError as e:
        raised = str(e)
    save("dcnn_error", message=np.array(raised))


# ------------------------------------------------------------------ MFA
def gen_mfa():
    mf = load_by_path("ref_mfa", "ocean/models/components/multi_factor_attention.py")
    for tag, (td, sdim, ld, ad, heads, B) in {"tiny": (12, 10, 24, 16, 4, 5), "default": (128, 128, 768, 128, 8, 3)}.items():
        m = mf.MultiFactorAttention(td, sdim, ld, ad, heads, dropout=0.0)
        sd0 = synth.fill_state_dict(m.state_dict(), 200)
        m.load_state_dict(sd0)
        t, s, l = rnd((B, td), 1), rnd((B, sdim), 2), rnd((B, ld), 3)
        for v in (t, s, l):
            v.requires_grad_(True)
        gw = rnd((B, ad), 4)
        m.train()
        y = m(t, s, l)
        (y * gw).sum().backward()
        arrs = dict(t=t.detach().numpy(), s=s.detach().numpy(), l=l.detach().numpy(), gw=gw.numpy(),
                    y=y.detach().numpy(), dt=t.grad.numpy(), ds=s.grad.numpy(), dl=l.grad.numpy(),
                    heads=np.array(heads), meta_seed=np.array(200))
        none_grad = [k for k, p in m.named_parameters() if p.grad is None]
        arrs["none_grad"] = np.array(none_grad)
        if tag == "tiny":
            arrs.update(npd(sd0, "sd/"))
            arrs.update({"grad/" + k: p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None})
        else:
            arrs.update({"gdig/" + k: synth.digest(p.grad).numpy() for k, p in m.named_parameters()
                         if p.grad is not None})
        save("mfa_" + tag, **arrs)


# ------------------------------------------------------------------ fusion / InfoNCE
def gen_fusion():
    gf = load_by_path("ref_graph_fusion", "ocean/models/components/graph_fusion.py")
    # InfoNCE, both branches
    a, p, n = rnd((6, 16), 1), rnd((6, 16), 2), rnd((9, 16), 3)
    a.requires_grad_(True)
    p.requires_grad_(True)
    loss_fn = gf.InfoNCELoss(0.1)
    lb = loss_fn(a, p, n)
    lb.backward()
    arrs = dict(a=a.detach().numpy(), p=p.detach().numpy(), n=n.numpy(), loss_bank=lb.detach().numpy(),
                da_bank=a.grad.numpy().copy(), dp_bank=p.grad.numpy().copy())
    a.grad = None
    p.grad = None
    li = loss_fn(a, p)
    li.backward()
    arrs.update(loss_inbatch=li.detach().numpy(), da_inbatch=a.grad.numpy(), dp_inbatch=p.grad.numpy())
    arrs["loss_inbatch_t05"] = gf.InfoNCELoss(0.5)(a.detach(), p.detach()).numpy()
    save("infonce", **arrs)

    for tag, (din, fd, pd_, B) in {"tiny": (16, 32, 16, 4), "default": (128, 256, 128, 3)}.items():
        m = gf.GraphFusionModule(din, din, din, fd, pd_, 0.1, "attention", dropout=0.0)
        zero_dropout(m)
        sd0 = synth.fill_state_dict(m.state_dict(), 300)
        m.load_state_dict(sd0)
        t, s = rnd((B, din), 1), rnd((B, din), 2)
        t.requires_grad_(True)
        s.requires_grad_(True)
        gw = rnd((B, fd), 4)
        m.train()
        torch.manual_seed(0)
        r = m(t, s, t, compute_contrastive_loss=True)
        closs = sum(r["contrastive_losses"].values())
        ((r["fused_representation"] * gw).sum() + 0.1 * closs).backward()
        arrs = dict(t=t.detach().numpy(), s=s.detach().numpy(), gw=gw.numpy(),
                    fused=r["fused_representation"].detach().numpy(), dt=t.grad.numpy(), ds=s.grad.numpy(),
                    meta_seed=np.array(300))
        for k, v in r["contrastive_losses"].items():
            arrs["closs/" + k] = v.detach().numpy()
        for k, v in r["projections"].items():
            arrs["proj/" + k] = v.detach().numpy()
        arrs["none_grad"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
        arrs["bank_after"] = m.contrastive_module.memory_bank.numpy().copy()
        arrs["ptr_after"] = m.contrastive_module.memory_ptr.numpy().copy()
        if tag == "tiny":
            arrs.update(npd(sd0, "sd/"))
            arrs.update({"grad/" + k: p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None})
        else:
            arrs.update({"gdig/" + k: synth.digest(p.grad).numpy() for k, p in m.named_parameters()
                         if p.grad is not None})
        m.eval()
        with torch.no_grad():
            arrs["fused_eval"] = m(t, s, t, compute_contrastive_loss=False)["fused_representation"].numpy()
        save("fusion_" + tag, **arrs)

    # memory-bank pointer/contents traces on the steps the reference survives
    for B, steps in {1: 12, 2: 10, 3: 8, 4: 8, 5: 8, 6: 8, 8: 2, 10: 1}.items():
        cm = gf.ContrastiveLearningModule({"temporal": 8, "spatial": 8, "log": 8}, 4, 0.1)
        cm.train()
        bank0 = cm.memory_bank.numpy().copy()
        ptrs, banks, fed = [], [], []
        for st in range(steps):
            rows = {k: rnd((B, 4), 1000 * B + 10 * st + i) for i, k in enumerate(("temporal", "spatial", "log"))}
            try:
                cm._update_memory_bank(rows)
            except RuntimeError:
                break
            fed.append(torch.cat(list(rows.values())).numpy())
            ptrs.append(int(cm.memory_ptr))
            banks.append(cm.memory_bank.numpy().copy())
        save(f"bank_trace_B{B}", ptrs=np.array(ptrs), banks=np.stack(banks), fed=np.stack(fed),
             bank0=bank0)


# ------------------------------------------------------------------ full model
def tiny_config(ocean):
    cfg = ocean.OCEANConfig()
    cfg.model.dcnn_channels = [8, 16]
    cfg.model.dcnn_dilation_rates = [1, 2]
    cfg.model.hidden_dim = 16
    cfg.model.attention_dim = 16
    cfg.model.attention_num_heads = 4
    cfg.model.gnn_num_heads = 2
    cfg.model.fusion_projection_dim = 32
    cfg.model.num_services = 6
    cfg.data.metrics_features = 5
    cfg.data.log_embedding_dim = 24
    cfg.data.graph_features = 7
    return cfg


def gen_model():
    pyg_shim.install()
    logging.disable(logging.CRITICAL)
    sys.path.insert(0, REF)
    import ocean  # the reference package
    from ocean.data.data_types import DatasetSample, ServiceGraph
    from ocean.models.training import AdaptiveLearningRateScheduler, LossFunction, OnlineLearner
    from datetime import datetime

    # G1: README example (README.md:140-171): eval, B=1, T=10, identity adjacency
    cfg = ocean.default_config()
    torch.manual_seed(42)
    m = ocean.OCEANModel(cfg)
    zero_dropout(m)
    sd0 = synth.fill_state_dict(m.state_dict(), 42)
    m.load_state_dict(sd0)
    keys = list(m.state_dict().keys())
    metrics, logs = rnd((1, 10, 12), 1), rnd((1, 10, 768), 2)
    g = ServiceGraph(adjacency_matrix=torch.eye(5), node_features=rnd((5, 12), 3),
                     service_names=[f"service_{i}" for i in range(5)])
    m.eval()
    with torch.no_grad():
        o = m(metrics, g, logs, return_intermediate=True)
    it = o["intermediate"]
    top = torch.topk(o["root_cause_probs"], 10, dim=-1)
    save("model_readme", metrics=metrics.numpy(), logs=logs.numpy(), adj=g.adjacency_matrix.numpy(),
         node_features=g.node_features.numpy(), probs=o["root_cause_probs"].numpy(),
         logits=o["root_cause_logits"].numpy(), temporal=it["temporal_features"].numpy(),
         spatial=it["spatial_features"].numpy(), attended=it["attended_features"].numpy(),
         fused=it["fused_representation"].numpy(), top10=top.indices.numpy(), meta_seed=np.array(42),
         state_keys=np.array(keys), state_shapes=np.array([str(tuple(v.shape)) for v in m.state_dict().values()]),
         param_names=np.array([k for k, _ in m.named_parameters()]))

    # G2: training steps.  tiny (full grads) and default architecture (digests)
    for tag, cfg, B, T, N in (("tiny", tiny_config(ocean), 4, 16, 6), ("default", ocean.default_config(), 4, 12, 6)):
        for neg in ("bank", "in_batch"):
            m = ocean.OCEANModel(cfg)
            zero_dropout(m)
            sd0 = synth.fill_state_dict(m.state_dict(), 7)
            m.load_state_dict(sd0)
            cm = m.fusion_module.contrastive_module
            if neg == "in_batch":
                cm._get_negatives = lambda: None
                cm._update_memory_bank = lambda projections: None
            F_, L_, G_ = m.metrics_input_dim, m.log_input_dim, m.graph_input_dim
            S = cfg.model.num_services
            adj = synth.random_digraph(N, 2.0, 11, self_loop_at=2, isolated=4)
            g = ServiceGraph(adjacency_matrix=adj, node_features=rnd((N, G_), 3))
            opt = torch.optim.Adam(m.parameters(), lr=cfg.training.learning_rate,
                                   weight_decay=cfg.training.weight_decay)
            arrs = dict(adj=adj.numpy(), node_features=g.node_features.numpy(), meta_seed=np.array(7),
                        meta_negatives=np.array(neg), meta_patch=np.array(
                            "dropout p=0" + ("; _get_negatives->None; bank update no-op" if neg == "in_batch" else "")))
            if tag == "tiny":
                arrs.update(npd(sd0, "sd/"))
            for step in range(3):
                metrics, logs = rnd((B, T, F_), 10 + step), rnd((B, L_), 20 + step)
                labels = torch.zeros(B, S)
                labels[torch.arange(B), torch.randint(0, S, (B,), generator=torch.Generator().manual_seed(30 + step))] = 1
                m.train()
                out = m(metrics, g, logs, return_intermediate=True)
                losses = m.compute_loss(out, labels)
                opt.zero_grad()
                losses["total_loss"].backward()
                norm = torch.nn.utils.clip_grad_norm_(m.parameters(), cfg.training.gradient_clip_norm)
                pre = f"s{step}/"
                arrs.update({pre + "metrics": metrics.numpy(), pre + "logs": logs.numpy(), pre + "labels": labels.numpy(),
                             pre + "probs": out["root_cause_probs"].detach().numpy(),
                             pre + "temporal": out["intermediate"]["temporal_features"].detach().numpy(),
                             pre + "spatial": out["intermediate"]["spatial_features"].detach().numpy(),
                             pre + "attended": out["intermediate"]["attended_features"].detach().numpy(),
                             pre + "fused": out["intermediate"]["fused_representation"].detach().numpy(),
                             pre + "grad_norm": norm.numpy()})
                for k, v in losses.items():
                    arrs[pre + "loss/" + k] = v.detach().numpy()
                if step == 0:
                    arrs["none_grad"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
                    if neg == "bank":   # LossFunction (trainer.py:59-120) on the same outputs
                        lf = LossFunction()
                        tl = lf(out, labels, out["intermediate"]["temporal_features"],
                                out["intermediate"]["temporal_features"].detach() * 0.9)
                        for k, v in tl.items():
                            arrs["trainer_loss/" + k] = torch.as_tensor(v).detach().numpy()
                    for k, p in m.named_parameters():
                        if p.grad is not None:
                            if tag == "tiny":
                                arrs["s0/cgrad/" + k] = p.grad.numpy().copy()       # clipped grads
                            else:
                                arrs["s0/cgdig/" + k] = synth.digest(p.grad).numpy()
                                arrs["s0/cgsamp/" + k] = synth.samples(p.grad).numpy()
                opt.step()
                if step in (0, 2):
                    for k, v in m.state_dict().items():
                        if tag == "tiny":
                            arrs[pre + "sd_after/" + k] = v.numpy().copy()
                        else:
                            arrs[pre + "sddig_after/" + k] = synth.digest(v).numpy()
                            if v.is_floating_point():
                                arrs[pre + "sdsamp_after/" + k] = synth.samples(v).numpy()
                arrs[pre + "bank"] = cm.memory_bank.numpy().copy()
                arrs[pre + "ptr"] = cm.memory_ptr.numpy().copy()
            save(f"model_train_{tag}_{neg}", **arrs)

    # G4: OnlineLearner return dicts (tiny config) and the LR scheduler trace
    cfg = tiny_config(ocean)
    m = ocean.OCEANModel(cfg)
    zero_dropout(m)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 9))
    learner = OnlineLearner(m, cfg)
    adj = synth.random_digraph(6, 2.0, 12)
    g = ServiceGraph(adjacency_matrix=adj, node_features=rnd((6, 7), 3))
    samples = []
    for i in range(6):
        lab = torch.zeros(6)
        lab[i % 6] = 1
        samples.append(DatasetSample(metrics=rnd((16, 5), 40 + i), logs=rnd((24,), 50 + i), graph=g,
                                     label=lab, timestamp=datetime(2024, 1, 1)))
    arrs = dict(adj=adj.numpy(), node_features=g.node_features.numpy(), meta_seed=np.array(9))
    for i in range(3):
        r = learner.process_sample(samples[i])
        for k, v in r.items():
            arrs[f"sample{i}/{k}"] = np.array(v)
    r = learner.process_batch(samples[3:6])
    for k, v in r.items():
        arrs[f"batch/{k}"] = np.array(v)
    for i, s_ in enumerate(samples):
        arrs[f"in/metrics{i}"], arrs[f"in/logs{i}"], arrs[f"in/label{i}"] = s_.metrics.numpy(), s_.logs.numpy(), s_.label.numpy()
    save("online_learner", **arrs)

    sched = AdaptiveLearningRateScheduler(initial_lr=1e-3)
    popt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    gl = torch.Generator().manual_seed(5)
    seq = [float(1.0 + 0.3 * torch.randn(1, generator=gl) - 0.002 * i) for i in range(150)] + \
          [float(0.7 + 0.01 * i) for i in range(60)]
    lrs = [sched.step(v, popt) for v in seq]
    save("lr_trace", losses=np.array(seq), lrs=np.array(lrs))


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["dcnn", "mfa", "fusion", "model"]
    for w in which:
        {"dcnn": gen_dcnn, "mfa": gen_mfa, "fusion": gen_fusion, "model": gen_model}[w]()
