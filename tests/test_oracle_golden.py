"""The CPU oracle (oracle/ocean_ref.py) against the fixtures the reference's
own classes produced (oracle/gen_golden.py).  fp32, dropout p=0.
Tolerances: values 2e-5 abs+rel (two fp32 evaluation orders of the same
formula), gradients 1e-4."""
import numpy as np
import pytest
import torch

from oracle import ocean_ref as R
from oracle import synth


def T(a):
    return torch.from_numpy(np.asarray(a))


def sd_of(g, prefix="sd/"):
    return {k[len(prefix):]: T(g[k]).clone() for k in g.files if k.startswith(prefix)}


def close(a, b, tol=2e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(b.abs().max().item(), 1.0) if b.numel() else 1.0
    assert err <= tol * scale, f"max err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("tag", ["tiny", "three", "t2"])
def test_dcnn_train_eval_grads(golden, tag):
    g = golden("dcnn_" + tag)
    sd = R.require_grad(sd_of(g))
    x = T(g["x"]).clone().requires_grad_(True)
    dil = [int(d) for d in g["dilations"]]
    y = R.dcnn_forward(sd, "", x, dil, training=True)
    close(y, g["y_train"])
    (y * T(g["gw"])).sum().backward()
    close(x.grad, g["dx"], 1e-4)
    for k in g.files:
        if k.startswith("grad/"):
            close(sd[k[5:]].grad, g[k], 1e-4)
        if k.startswith("sd_after/") and ("running" in k or "tracked" in k):
            close(sd[k[9:]], g[k])
    with torch.no_grad():
        close(R.dcnn_forward(sd, "", x, dil, training=False), g["y_eval"])


def dcnn_default_sd(seed):
    import torch.nn as nn
    shapes = {}
    ch = [64, 128, 256, 512]
    shapes["input_projection.weight"] = (64, 12)
    shapes["input_projection.bias"] = (64,)
    for i, c in enumerate(ch):
        shapes[f"conv_layers.{i}.conv.weight"] = (c, c, 3)
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"conv_layers.{i}.batch_norm.{n}"] = (c,)
        shapes[f"conv_layers.{i}.batch_norm.num_batches_tracked"] = ()
    for i in range(3):
        shapes[f"transition_layers.{i}.0.weight"] = (ch[i + 1], ch[i], 1)
        for n in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"transition_layers.{i}.1.{n}"] = (ch[i + 1],)
        shapes[f"transition_layers.{i}.1.num_batches_tracked"] = ()
    shapes["output_projection.weight"] = (128, 512)
    shapes["output_projection.bias"] = (128,)
    sd = {k: (torch.zeros(s, dtype=torch.long) if k.endswith("tracked") else torch.zeros(s)) for k, s in shapes.items()}
    return synth.fill_state_dict(sd, seed)


def test_dcnn_default_architecture(golden):
    """Key order of the state_dict matters for the seeded fill: it must be the
    reference module's registration order."""
    g = golden("dcnn_default")
    sd = R.require_grad(dcnn_default_sd(int(g["meta_seed"])))
    x = T(g["x"])
    y = R.dcnn_forward(sd, "", x, [1, 2, 4, 8], training=True)
    close(y, g["y_train"], 5e-5)
    (y * T(g["gw"])).sum().backward()
    for k in g.files:
        if k.startswith("gdig/"):
            close(synth.digest(sd[k[5:]].grad), g[k], 2e-4)
        if k.startswith("sd_after/"):
            close(sd[k[9:]], g[k])
    with torch.no_grad():
        close(R.dcnn_forward(sd, "", x, [1, 2, 4, 8], training=False), g["y_eval"], 5e-5)


def test_dcnn_single_value_per_channel_raises(golden):
    msg = str(golden("dcnn_error")["message"])
    assert "Expected more than 1 value per channel" in msg
    sd = dcnn_default_sd(1)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        R.dcnn_forward(sd, "", torch.randn(1, 1, 12), [1, 2, 4, 8], training=True)


def test_mfa_tiny(golden):
    g = golden("mfa_tiny")
    sd = R.require_grad(sd_of(g))
    t, s, l = (T(g[k]).clone().requires_grad_(True) for k in ("t", "s", "l"))
    y = R.mfa_forward(sd, "", t, s, l)
    close(y, g["y"])
    (y * T(g["gw"])).sum().backward()
    close(t.grad, g["dt"], 1e-4)
    close(s.grad, g["ds"], 1e-4)
    close(l.grad, g["dl"], 1e-4)
    none = set(str(k) for k in g["none_grad"])
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            if ".w_q." in name or ".w_k." in name:
                # exactly-zero gradients in the reference; the restatement
                # drops the dead projections so they get no gradient at all
                assert np.all(g[k] == 0)
                continue
            close(sd[name].grad, g[k], 1e-4)
    for k, v in sd.items():
        if k in none:
            assert v.grad is None
    assert all("_attention.multihead" in k or "_attention.layer_norm" in k or "_attention.feed_forward" in k
               or "_attention.ff_layer_norm" in k for k in none)


def test_infonce_both_branches(golden):
    g = golden("infonce")
    for branch, neg in (("bank", T(g["n"])), ("inbatch", None)):
        a = T(g["a"]).clone().requires_grad_(True)
        p = T(g["p"]).clone().requires_grad_(True)
        loss = R.infonce(a, p, neg, 0.1)
        close(loss, g["loss_" + branch])
        loss.backward()
        close(a.grad, g["da_" + branch], 1e-4)
        close(p.grad, g["dp_" + branch], 1e-4)
    close(R.infonce(T(g["a"]), T(g["p"]), None, 0.5), g["loss_inbatch_t05"])


def test_fusion_tiny(golden):
    g = golden("fusion_tiny")
    sd = R.require_grad(sd_of(g))
    t = T(g["t"]).clone().requires_grad_(True)
    s = T(g["s"]).clone().requires_grad_(True)
    r = R.fusion_forward(sd, "", t, s, t, training=True)
    close(r["fused_representation"], g["fused"])
    names = ("temporal_spatial_contrastive", "temporal_log_contrastive", "spatial_log_contrastive")
    for k in names:
        close(r[k], g["closs/" + k])
    for z, k in zip(r["projections"], ("temporal", "spatial", "log")):
        close(z, g["proj/" + k])
    ((r["fused_representation"] * T(g["gw"])).sum() + 0.1 * sum(r[k] for k in names)).backward()
    close(t.grad, g["dt"], 1e-4)
    close(s.grad, g["ds"], 1e-4)
    for k in g.files:
        if k.startswith("grad/"):
            close(sd[k[5:]].grad, g[k], 1e-4)
    assert set(str(k) for k in g["none_grad"]) == {k for k in R.trainable_keys(sd) if sd[k].grad is None}
    close(sd["contrastive_module.memory_bank"], g["bank_after"])
    assert int(sd["contrastive_module.memory_ptr"]) == int(g["ptr_after"][0])
    with torch.no_grad():
        close(R.fusion_forward(sd, "", t, s, t, training=False)["fused_representation"], g["fused_eval"])


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 6, 8, 10])
def test_memory_bank_ring_matches_reference_while_it_survives(golden, B):
    g = golden(f"bank_trace_B{B}")
    bank = T(g["bank0"]).clone()
    ptr = 0
    assert len(g["ptrs"]) >= 1
    for st in range(len(g["ptrs"])):
        ptr = R.memory_bank_update(bank, ptr, T(g["fed"][st]))
        assert ptr == int(g["ptrs"][st]), (st, ptr)
        close(bank, g["banks"][st], 0)
    # beyond the reference's survival the ring keeps going without raising
    for _ in range(5):
        ptr = R.memory_bank_update(bank, ptr, torch.randn(3 * B, bank.shape[1]))
        assert 0 <= ptr <= bank.shape[0]


def test_lr_scheduler_trace(golden):
    g = golden("lr_trace")
    got = R.adaptive_lr_trace([float(v) for v in g["losses"]])
    np.testing.assert_allclose(np.array(got), g["lrs"], rtol=1e-12)
    assert got[-1] < got[0]
