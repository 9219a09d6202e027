"""Device timing of the conv-shaped GEMMs of the dilated CNN (development aid).
usage: python tools/bench_gemm.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean import _hip
if os.environ.get("ZERO"):
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)

B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), (int(sys.argv[2]) if len(sys.argv) > 2 else 128)
M = B * T
L = _hip.lib()
st = lambda: _hip.stream()


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = []
for C, d in ((64, 1), (128, 2), (256, 4), (512, 8)):
    x = torch.randn(M, C, device="cuda")
    w = torch.randn(3, C, C, device="cuda") * 0.05
    y = torch.empty(M, C, device="cuda")
    dw = torch.zeros(3, C, C, device="cuda")
    fl = 2.0 * M * C * C * 3
    t = timeit(lambda: L.ocean_conv_gemm_f32(x.data_ptr(), C, w.data_ptr(), C, y.data_ptr(), C, M, C, C, 0, 0, 3, T, -d, d, C * C, 0, 1, st()))
    rows.append((f"conv fwd  C={C}", t, fl))
    t = timeit(lambda: L.ocean_conv_gemm_f32(y.data_ptr(), C, w.data_ptr(), C, x.data_ptr(), C, M, C, C, 0, 1, 3, T, d, -d, C * C, 0, 1, st()))
    rows.append((f"conv dgrad C={C}", t, fl))
    t = timeit(lambda: L.ocean_conv_gemm_f32(y.data_ptr(), C, x.data_ptr(), C, dw.data_ptr(), C, C, C, M, 1, 1, 3, T, -d, d, 0, C * C, 0, st()))
    rows.append((f"conv wgrad C={C}", t, fl))
for Ci, Co in ((64, 128), (128, 256), (256, 512)):
    x = torch.randn(M, Ci, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") * 0.05
    y = torch.empty(M, Co, device="cuda")
    dw = torch.zeros(Co, Ci, device="cuda")
    fl = 2.0 * M * Ci * Co
    rows.append((f"1x1 fwd  {Ci}->{Co}", timeit(lambda: _hip.gemm(x, w, y, M, Co, Ci)), fl))
    rows.append((f"1x1 dgrad {Ci}->{Co}", timeit(lambda: _hip.gemm(y, w, x, M, Ci, Co, b_kn=True)), fl))
    rows.append((f"1x1 wgrad {Ci}->{Co}", timeit(lambda: _hip.gemm(y, x, dw, Co, Ci, M, a_kmajor=True, b_kn=True)), fl))
tot_t = tot_f = 0.0
for name, t, fl in rows:
    print(f"{name:22s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s")
    tot_t += t; tot_f += fl
print(f"{'TOTAL':22s} {tot_t:8.3f} ms  {tot_f / tot_t / 1e9:7.1f} TFLOP/s  (env OCEAN_GEMM_BK={os.environ.get('OCEAN_GEMM_BK', '')})")

# the small linear layers of the attention / fusion / head stages: rows = batch (or 3 x batch)
rows = []
for Mr, K, N in ((B, 768, 128), (B, 128, 128), (B, 384, 128), (3 * B, 128, 384), (3 * B, 128, 64), (B, 128, 64), (B, 256, 128), (B, 64, 256)):
    x = torch.randn(Mr, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    y = torch.empty(Mr, N, device="cuda")
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * Mr * K * N
    rows.append((f"lin fwd  {Mr}x{K}->{N}", timeit(lambda: _hip.gemm(x, w, y, Mr, N, K), 20), fl))
    rows.append((f"lin dgrad {Mr}x{K}->{N}", timeit(lambda: _hip.gemm(y, w, x, Mr, K, N, b_kn=True), 20), fl))
    rows.append((f"lin wgrad {Mr}x{K}->{N}", timeit(lambda: _hip.gemm(y, x, dw, N, K, Mr, a_kmajor=True, b_kn=True), 20), fl))
tot_t = tot_f = 0.0
for name, t, fl in rows:
    print(f"{name:22s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s")
    tot_t += t; tot_f += fl
print(f"{'TOTAL':22s} {tot_t:8.3f} ms  {tot_f / tot_t / 1e9:7.1f} TFLOP/s  (env OCEAN_GEMM_BK={os.environ.get('OCEAN_GEMM_BK', '')})")
