"""Fixed per-tile cost vs loop cost of the 128x128 GEMM tile: plain NN / NT GEMMs with M = 131072, N = 256 or 512 and a
swept reduction length K; prints time, rate and the least-squares fit time = a + b*K (a = what a launch spends outside
the k-loop, summed over its tile rounds).  usage: python tools/bench_gemm_ksweep.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean import _hip

M = 131072


def timeit(fn, n=6):
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


for N in (256, 512):
    for mode in ("fwd (NT)", "dgrad (NN)"):
        pts = []
        for K in (64, 128, 256, 512, 768, 1024, 1536, 2048):
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") * 0.05 if mode.startswith("fwd") else torch.randn(K, N, device="cuda") * 0.05
            y = torch.empty(M, N, device="cuda")
            t = timeit((lambda: _hip.gemm(x, w, y, M, N, K)) if mode.startswith("fwd") else (lambda: _hip.gemm(x, w, y, M, N, K, b_kn=True)))
            pts.append((K, t))
            print(f"N={N} {mode:10s} K={K:5d}  {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:6.1f} TFLOP/s", flush=True)
        n = len(pts); sx = sum(k for k, _ in pts); sy = sum(t for _, t in pts)
        sxx = sum(k * k for k, _ in pts); sxy = sum(k * t for k, t in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"N={N} {mode}: fit  {a * 1e3:.1f} us + {b * 1e3:.4f} us/K  -> loop rate {2.0 * M * N / b / 1e9:.1f} TFLOP/s, "
              f"fixed part = {a / b:.0f} K-equivalents", flush=True)
