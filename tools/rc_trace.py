"""Per-operation timeline inside the stage kernels (attention / fusion / head chains of small batches): runs a few update
steps eagerly with the library's diagnostic trace buffer set and prints, for the last forward / backward kernels, the
time block 0 spent in each operation.  usage: python tools/rc_trace.py [BATCH] [SEQ] [SERVICES]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
import ocean
from ocean import _hip
from ocean.data import ServiceGraph
from ocean.models.training import OnlineLearner

B, T, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 64), (int(sys.argv[3]) if len(sys.argv) > 3 else 64)
KINDS = ["LIN", "LN_FWD", "LN_BWD", "L2N_FWD", "L2N_BWD", "CROSS_FWD", "CROSS_BWD", "GATE_FWD", "GATE_BWD", "MHA3_FWD", "MHA3_BWD", "MEAN_MID",
         "SIGMOID", "SIGMOID_BWD", "RELU_BWD", "COPY2D"]
cfg = ocean.default_config(); cfg.model.num_services = N
torch.manual_seed(0)
m = ocean.OCEANModel(cfg).cuda(); m.contrastive_negatives = "in_batch"
lr = OnlineLearner(m, cfg)
g = torch.Generator().manual_seed(1)
x, lg = torch.randn(B, T, 12, generator=g).cuda(), torch.randn(B, 768, generator=g).cuda()
y = torch.zeros(B, N).cuda(); y[:, 0] = 1
sg = ServiceGraph((torch.rand(N, N, generator=g) < 4.0 / N).float(), torch.randn(N, 12, generator=g))
for _ in range(3):
    lr.process_tensors(x, sg, lg, y)
torch.cuda.synchronize()
L = _hip.lib()
# one buffer per stage call: hook by running each stage separately is intrusive; instead trace the whole step and keep
# only the LAST kernel's record - so run the step with the buffer set and print after each of several partial runs
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
m.train()


def show(tag):
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    rows, n = [], 0          # (kind, stamp, operations sharing the stamp: the members of a group of independent layers)
    for i in range(0, 60):
        k, c = t[2 * i], t[2 * i + 1]
        n = i
        if k == -1:
            rows.append((k, c, 0))
            break
        if c == 0 and rows:      # a group member: it runs inside its group head's pass
            rows[-1] = (rows[-1][0], rows[-1][1], rows[-1][2] + 1)
        else:
            rows.append((k, c, 1))
    cyc, wall = t[2 * n + 3] - t[2 * n + 2], (t[2 * n + 1] - t[2 * n + 4]) / 100.0
    print(f"== {tag}: {n} operations, {(rows[-1][1] - rows[0][1]) / 100.0:.1f} us; whole kernel {wall:.1f} us = {cyc} shader cycles -> {cyc / max(wall, 1e-9) / 1e3:.2f} GHz")
    for (k, c, members), (_, c2, _m) in zip(rows[:-1], rows[1:]):
        group = f"  (group of {members} independent layers)" if members > 1 else ""
        print(f"   {KINDS[k // 1000]:12s} width {k % 1000:4d}  {(c2 - c) / 100.0:6.2f} us{group}")
    buf.zero_()


L.ocean_rc_debug_trace(buf.data_ptr())
t_feat = torch.randn(B, 128, generator=g).cuda().requires_grad_(True)
s_feat = torch.randn(B, 128, generator=g).cuda().requires_grad_(True)
att = m.attention_module(t_feat, s_feat, lg)
show("attention forward")
att.sum().backward()
show("attention backward (last chain of it)")
fm = m.fusion_module
out = fm(att.detach().requires_grad_(True), s_feat.detach(), att.detach(), compute_contrastive_loss=True)
torch.cuda.synchronize()
show("fusion forward (last chain of it)")
L.ocean_rc_debug_trace(None)
