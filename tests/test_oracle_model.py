"""Whole-model oracle (forward, losses, clip + Adam steps, memory bank) against
fixtures produced by the reference's OCEANModel / OnlineLearner / LossFunction
(oracle/gen_golden.py; GATConv through the restated shim => the GNN numbers
are pinned only up to that shim)."""
import ast

import numpy as np
import pytest
import torch

from oracle import ocean_ref as R
from oracle import synth
from test_oracle_golden import T, close, sd_of


def default_sd(golden, seed):
    g = golden("model_readme")
    sd = {}
    for k, s in zip(g["state_keys"], g["state_shapes"]):
        shape = ast.literal_eval(str(s))
        k = str(k)
        sd[k] = torch.zeros(shape, dtype=torch.long) if k.endswith(("tracked", "memory_ptr")) else torch.zeros(shape)
    return synth.fill_state_dict(sd, seed)


TINY = dict(dilations=(1, 2), gnn_heads=2, gnn_layers=2, fusion_heads=8, log_dim=24)
DEFAULT = dict(dilations=(1, 2, 4, 8), gnn_heads=4, gnn_layers=2, fusion_heads=8, log_dim=768)


def test_readme_example_eval(golden):
    g = golden("model_readme")
    sd = default_sd(golden, int(g["meta_seed"]))
    with torch.no_grad():
        o = R.model_forward(sd, T(g["metrics"]), T(g["adj"]), T(g["node_features"]), T(g["logs"]),
                            training=False, **DEFAULT)
    for k, name in (("temporal_features", "temporal"), ("spatial_features", "spatial"),
                    ("attended_features", "attended"), ("fused_representation", "fused"),
                    ("root_cause_logits", "logits")):
        close(o[k], g[name], 5e-5)
    rel = ((o["root_cause_probs"] - T(g["probs"])).abs() / T(g["probs"])).max().item()
    assert rel < 1e-4
    assert torch.equal(torch.topk(o["root_cause_probs"], 10, dim=-1).indices, T(g["top10"]))


def test_gnn_dedup_equals_per_replica():
    sd = {k[len("spatial_encoder."):]: v for k, v in default_sd_cache().items() if k.startswith("spatial_encoder.")}
    adj = synth.random_digraph(9, 3.0, 5, self_loop_at=1, isolated=7)
    nf = torch.randn(9, 12, generator=torch.Generator().manual_seed(1))
    ei = R.edge_index_of(adj)
    a = R.gnn_forward(sd, "", nf, ei, 3, 4)
    b = R.gnn_forward_dedup(sd, "", nf, ei, 3, 4)
    close(a, b, 1e-6)


_cache = {}


def default_sd_cache():
    if not _cache:
        import os
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_readme.npz"))
        sd = {}
        for k, s in zip(g["state_keys"], g["state_shapes"]):
            k = str(k)
            shape = ast.literal_eval(str(s))
            sd[k] = torch.zeros(shape, dtype=torch.long) if k.endswith(("tracked", "memory_ptr")) else torch.zeros(shape)
        _cache.update(synth.fill_state_dict(sd, 3))
    return _cache


def run_steps(g, sd, arch, neg, tiny):
    """Re-run the 3 recorded update steps with the oracle's own clip + Adam."""
    sd = R.require_grad(sd)
    tk = [k for k in R.trainable_keys(sd) if sd[k].is_floating_point()]
    state = {k: {} for k in tk}
    adj, nf = T(g["adj"]), T(g["node_features"])
    for step in range(3):
        pre = f"s{step}/"
        for k in tk:
            sd[k].grad = None
        out = R.model_forward(sd, T(g[pre + "metrics"]), adj, nf, T(g[pre + "logs"]), training=True,
                              negatives=neg, **arch)
        losses = R.compute_loss(out, T(g[pre + "labels"]))
        rel = ((out["root_cause_probs"] - T(g[pre + "probs"])).abs() / T(g[pre + "probs"])).max().item()
        assert rel < 1e-4, (step, rel)
        for k in ("temporal", "spatial", "attended", "fused"):
            close(out[k + "_features" if k != "fused" else "fused_representation"], g[pre + k], 1e-4)
        for k in g.files:
            if k.startswith(pre + "loss/"):
                close(losses[k[len(pre) + 5:]], g[k], 1e-4)
        if step == 0 and neg == "bank":
            tl = R.trainer_loss(out, T(g[pre + "labels"]), out["temporal_features"].detach() * 0.9)
            for k in g.files:
                if k.startswith("trainer_loss/"):
                    close(torch.as_tensor(tl[k[13:]]), g[k], 1e-4)
        losses["total_loss"].backward()
        with_grad = [k for k in tk if sd[k].grad is not None]
        grads = [sd[k].grad for k in with_grad]
        norm = R.clip_grad_norm(grads, 1.0)
        assert abs(norm - float(g[pre + "grad_norm"])) <= 1e-3 * float(g[pre + "grad_norm"])
        if step == 0:
            ref_none = set(str(k) for k in g["none_grad"])
            zero_in_ref = {k for k in tk if (".w_q." in k or ".w_k." in k) and "_self_attention" in k}
            assert {k for k in tk if sd[k].grad is None} == ref_none | zero_in_ref
            for k in with_grad:
                if tiny:
                    close(sd[k].grad, g["s0/cgrad/" + k], 1e-3)
                else:
                    close(synth.digest(sd[k].grad), g["s0/cgdig/" + k], 1e-3)
                    ref = T(g["s0/cgsamp/" + k])          # 256 evenly spaced elements, each on its own
                    err = (synth.samples(sd[k].grad) - ref).abs()
                    assert bool((err <= 1e-3 * ref.abs() + 1e-5 * max(float(ref.abs().max()), 1e-6)).all()), (k, float(err.max()))
        # the 12 dead q/k tensors have exactly-zero grads in the reference and
        # therefore still take Adam's coupled weight decay
        for k in tk:
            if sd[k].grad is None and ((".w_q." in k or ".w_k." in k) and "_self_attention" in k):
                sd[k].grad = torch.zeros_like(sd[k])
        with_grad = [k for k in tk if sd[k].grad is not None]
        with torch.no_grad():
            R.adam_step([sd[k] for k in with_grad], [sd[k].grad for k in with_grad], [state[k] for k in with_grad])
        close(sd["fusion_module.contrastive_module.memory_bank"], g[pre + "bank"], 1e-5)
        assert int(sd["fusion_module.contrastive_module.memory_ptr"]) == int(g[pre + "ptr"][0])
        if step in (0, 2):
            for k in sd:
                if tiny:
                    close(sd[k].detach(), g[pre + "sd_after/" + k], 2e-4)
                else:
                    # Adam turns fp32 noise on near-zero gradients into +-lr
                    # moves, so sums over big tensors only agree to ~1e-3
                    close(synth.digest(sd[k].detach()), g[pre + "sddig_after/" + k], 2e-3)
                    if sd[k].is_floating_point():
                        ref = T(g[pre + "sdsamp_after/" + k])
                        err = (synth.samples(sd[k].detach()) - ref).abs()
                        assert bool((err <= 1e-3 * ref.abs() + 2e-3 * (step + 1)).all()), (k, float(err.max()))
    return sd


@pytest.mark.parametrize("neg", ["bank", "in_batch"])
def test_three_update_steps_tiny(golden, neg):
    g = golden(f"model_train_tiny_{neg}")
    run_steps(g, sd_of(g), TINY, neg, tiny=True)


@pytest.mark.parametrize("neg", ["bank", "in_batch"])
def test_three_update_steps_default_arch(golden, neg):
    g = golden(f"model_train_default_{neg}")
    run_steps(g, default_sd(golden, int(g["meta_seed"])), DEFAULT, neg, tiny=False)
