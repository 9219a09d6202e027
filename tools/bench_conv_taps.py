"""Where a dilated-conv data-gradient launch differs from a plain GEMM of the same size (development aid): C = 256 and
512, dilation swept, against the plain NN product with K = 3C.  usage: python tools/bench_conv_taps.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean import _hip

B, T = 1024, 128
M = B * T
L = _hip.lib()


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


for C in (256, 512):
    x = torch.randn(M, C, device="cuda")
    y = torch.randn(M, C, device="cuda")
    w = torch.randn(3, C, C, device="cuda") * 0.05
    fl = 2.0 * M * C * C * 3
    for d in (0, 1, 2, 4, 8, 16):
        t = timeit(lambda: L.ocean_conv_gemm_f32(y.data_ptr(), C, w.data_ptr(), C, x.data_ptr(), C, M, C, C, 0, 1, 3, T, d, -d, C * C, 0, 1, _hip.stream()))
        print(f"C={C} conv dgrad taps=3 dilation={d:2d}  {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TFLOP/s", flush=True)
    xx = torch.randn(M, 3 * C, device="cuda")
    ww = torch.randn(3 * C, C, device="cuda") * 0.05
    t = timeit(lambda: _hip.gemm(xx, ww, y, M, C, 3 * C, b_kn=True))
    print(f"C={C} plain NN K={3 * C}            {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TFLOP/s", flush=True)
    t = timeit(lambda: _hip.gemm(x, w[0], y, M, C, C, b_kn=True))
    print(f"C={C} plain NN K={C} (one tap)      {t * 1e3:8.1f} us  {fl / 3 / t / 1e9:6.1f} TFLOP/s", flush=True)

# data dependence: the same launches on zeros, and on inputs whose every 8th row is zero
for C in (256,):
    w = torch.randn(3, C, C, device="cuda") * 0.05
    x = torch.empty(M, C, device="cuda")
    fl = 2.0 * M * C * C * 3
    for name, y in (("random", torch.randn(M, C, device="cuda")), ("zeros", torch.zeros(M, C, device="cuda")),
                    ("every 8th row zero", torch.randn(M, C, device="cuda") * (torch.arange(M, device="cuda") % 8 != 0).float()[:, None])):
        for d in (0, 16):
            t = timeit(lambda: L.ocean_conv_gemm_f32(y.data_ptr(), C, w.data_ptr(), C, x.data_ptr(), C, M, C, C, 0, 1, 3, T, d, -d, C * C, 0, 1, _hip.stream()))
            print(f"C={C} conv dgrad dilation={d:2d} input {name:20s} {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TFLOP/s", flush=True)
