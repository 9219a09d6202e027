"""Does a HIP graph run two independent small kernels of a captured step side by side?  40 few-tile GEMMs (a linear
layer's data gradient and weight gradient, 20 layers) captured on one stream, against the same 40 with each weight
gradient forked onto a second stream and joined back.  usage: python tools/try_graph_fork.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ocean-online-multi-modal-root-cause-analysis_amd"))
import torch
from ocean import _hip

B, N, K = 1024, 128, 256
dy = [torch.randn(B, N, device="cuda") for _ in range(20)]
x = [torch.randn(B, K, device="cuda") for _ in range(20)]
w = [torch.randn(N, K, device="cuda") for _ in range(20)]
dx = [torch.empty(B, K, device="cuda") for _ in range(20)]
dw = [torch.zeros(N, K, device="cuda") for _ in range(20)]


def layer(i, side=None):
    main = torch.cuda.current_stream()
    if side is not None:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _hip.gemm(dy[i], x[i], dw[i], N, K, B, a_kmajor=True, b_kn=True)
    else:
        _hip.gemm(dy[i], x[i], dw[i], N, K, B, a_kmajor=True, b_kn=True)
    _hip.gemm(dy[i], w[i], dx[i], B, K, N, b_kn=True)
    if side is not None:
        main.wait_stream(side)


def bench(side):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(20):
            layer(i, None)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(20):
                layer(i, side)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50 * 1e3


print("one stream      : %.1f us per 20 layers" % bench(None))
print("forked per layer: %.1f us per 20 layers" % bench(torch.cuda.Stream()))
