"""ocean_infonce_fwd_static - the InfoNCE forward whose workspace is an array built into the library - against
ocean_infonce_fwd on the same inputs, through the C ABI.  Two calls in a row share the array, the second one smaller
(stale contents of the first must not leak into it); both branches of the loss; sizes beyond the array are refused.
Same kernels on the same data: the loss values agree to 1e-6 relative (the row kernel adds per-block partial sums
with atomics, whose order may differ)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _shape(B, Bp, P, K, label_offset=0):
    from ocean import _hip
    sh = _hip.InfoNceShape()
    sh.B, sh.Bp, sh.P, sh.K, sh.label_offset, sh.mean_rows, sh.tau = B, Bp, P, K, label_offset, 0, 0.1
    return sh


def _both(B, Bp, P, K, seed, label_offset=0):
    """(loss through the caller-workspace entry point, loss through the built-in-scratch entry point)"""
    from ocean import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(B, P, generator=g).cuda()
    p = torch.randn(Bp, P, generator=g).cuda()
    n = torch.randn(K, P, generator=g).cuda() if K > 0 else None
    sh = _shape(B, Bp, P, K, label_offset)
    ws = torch.empty(L.ocean_infonce_workspace(ctypes.byref(sh)), device="cuda", dtype=torch.float32)
    ref, got = torch.zeros((), device="cuda"), torch.zeros((), device="cuda")
    _hip.check(L.ocean_infonce_fwd(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), _hip.dev_ptr(n), _hip.dev_ptr(ws), ws.numel(),
                                   _hip.dev_ptr(ref), _hip.stream()), "ocean_infonce_fwd")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    _hip.check(L.ocean_infonce_fwd_static(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), _hip.dev_ptr(n), _hip.dev_ptr(got),
                                          _hip.stream()), "ocean_infonce_fwd_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before          # the device's free memory did not move
    return float(ref), float(got)


def _check(ref, got):
    assert ref == ref and abs(got - ref) <= 1e-6 * abs(ref), (ref, got)


def test_in_batch_large_then_small():
    _check(*_both(1024, 1024, 128, 0, seed=1))                   # the headline size: 1.32 M floats of the 1.5 M array
    _check(*_both(37, 37, 24, 0, seed=2))                        # straight after, much smaller: nothing stale is read
    _check(*_both(6, 17, 24, 0, seed=3, label_offset=8))         # local rows against gathered positives


def test_memory_bank_branch_large_then_small():
    _check(*_both(512, 512, 128, 16, seed=4))
    _check(*_both(3, 3, 128, 16, seed=5))


def test_sizes_beyond_the_array_are_refused_and_loss_accumulates():
    from ocean import _hip
    L = _hip.lib()
    sh = _shape(2048, 2048, 128, 0)
    x = torch.randn(2048, 128, device="cuda")
    loss = torch.zeros((), device="cuda")
    st = L.ocean_infonce_fwd_static(ctypes.byref(sh), _hip.dev_ptr(x), _hip.dev_ptr(x), None, _hip.dev_ptr(loss), _hip.stream())
    assert st == 5
    with pytest.raises(_hip.OceanHipError, match="built-in scratch"):
        _hip.check(st, "ocean_infonce_fwd_static")
    assert float(loss) == 0.0
    # *loss is added to, like ocean_infonce_fwd's
    sh = _shape(8, 8, 16, 0)
    a, p = torch.randn(8, 16, device="cuda"), torch.randn(8, 16, device="cuda")
    for _ in range(2):
        _hip.check(L.ocean_infonce_fwd_static(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), None, _hip.dev_ptr(loss), _hip.stream()), "static")
    one = torch.zeros((), device="cuda")
    _hip.check(L.ocean_infonce_fwd_static(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), None, _hip.dev_ptr(one), _hip.stream()), "static")
    assert abs(float(loss) - 2 * float(one)) <= 1e-6 * abs(float(loss))


def test_recorded_into_a_hip_graph_after_a_warm_up():
    """The use the entry point exists for: after ocean_static_scratch_warmup() outside the capture, the call is recorded
    into a HIP graph (no allocation, no symbol lookup inside) and every replay adds the loss of the current inputs."""
    from ocean import _hip
    L = _hip.lib()
    _hip.check(L.ocean_static_scratch_warmup(), "ocean_static_scratch_warmup")
    B, P = 64, 32
    sh = _shape(B, B, P, 0)
    g = torch.Generator().manual_seed(11)
    a, p = torch.randn(B, P, generator=g).cuda(), torch.randn(B, P, generator=g).cuda()
    loss = torch.zeros((), device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with _hip.capture(graph):
        _hip.check(L.ocean_infonce_fwd_static(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), None, _hip.dev_ptr(loss), _hip.stream()),
                   "ocean_infonce_fwd_static under capture")
    ws = torch.empty(L.ocean_infonce_workspace(ctypes.byref(sh)), device="cuda", dtype=torch.float32)
    for step in range(2):
        ref = torch.zeros((), device="cuda")
        _hip.check(L.ocean_infonce_fwd(ctypes.byref(sh), _hip.dev_ptr(a), _hip.dev_ptr(p), None, _hip.dev_ptr(ws), ws.numel(),
                                       _hip.dev_ptr(ref), _hip.stream()), "ocean_infonce_fwd")
        loss.zero_()
        graph.replay()
        torch.cuda.synchronize()
        _check(float(ref), float(loss))
        a.copy_(torch.randn(B, P, generator=g))          # the next replay reads the new anchors in place
