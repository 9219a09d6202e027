"""Weight-gradient launches (gemm_tn_kernel) at the row counts of small per-GPU batches: time against the number of row
slices.  usage: python tools/tn_split_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean import _hip

L = _hip.lib()
T = 128


def run(Co, Ci, taps, K, split, reps=30):
    dy = torch.randn(K, Co, device="cuda"); x = torch.randn(K, Ci, device="cuda")
    dw = torch.zeros(taps, Co, Ci, device="cuda")
    def go():
        rc = L.ocean_conv_gemm_f32(dy.data_ptr(), Co, x.data_ptr(), Ci, dw.data_ptr(), Ci, Co, Ci, K, 1, 1, taps, T, -2 if taps == 3 else 0, 2 if taps == 3 else 0,
                                   0, Co * Ci, split, _hip.stream())
        assert rc == 0
    for _ in range(3):
        go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        go()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K in (16384, 32768, 65536):
    for Co, Ci, taps in ((128, 128, 3), (256, 256, 3), (512, 512, 3), (256, 128, 1), (512, 256, 1)):
        row = [f"K={K:6d} {Co:3d}x{Ci:3d} taps {taps}:"]
        for split in (0, 4, 8, 16, 24, 32, 48, 64, 90, 128):
            if split and K // split < 128:
                continue
            us = run(Co, Ci, taps, K, split)
            tf = 2.0 * Co * Ci * K * taps / us / 1e6
            row.append(f"{'auto' if split == 0 else split}:{us:6.1f}us/{tf:5.1f}TF")
        print("  ".join(row), flush=True)
