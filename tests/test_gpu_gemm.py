"""fp32 MFMA GEMM through the C ABI against torch fp32 matmul on the same
device data moved to the CPU (float64 reference).  Tolerance: fp32 GEMM with
K <= 2048 carries <= ~2e-6 relative error per element; 2e-5 of the result
scale is asserted."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _close(got, ref, tol=2e-5):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1.0)
    assert err <= tol * scale, f"max err {err:.3e} scale {scale:.3e}"


SHAPES = [(1, 1, 1), (3, 5, 7), (64, 64, 16), (65, 63, 17), (128, 128, 32), (130, 257, 100), (1024, 128, 768),
          (257, 100, 64), (4096, 64, 192), (33000, 128, 64), (40000, 512, 40), (7, 384, 12)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nt_linear_forward(M, N, K):
    from ocean import _hip
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    out = torch.full((M, N), float("nan"), device="cuda")
    _hip.gemm(a.cuda(), w.cuda(), out, M, N, K, bias=b.cuda())
    _close(out, a.double() @ w.double().t() + b.double())
    r = _rand(M, N, seed=4)
    out2 = torch.empty(M, N, device="cuda")
    _hip.gemm(a.cuda(), w.cuda(), out2, M, N, K, residual=r.cuda(), relu=True)
    _close(out2, torch.relu(a.double() @ w.double().t() + r.double()))


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nn_data_gradient(M, N, K):
    from ocean import _hip
    a, b = _rand(M, K, seed=1), _rand(K, N, seed=2)
    out = torch.empty(M, N, device="cuda")
    _hip.gemm(a.cuda(), b.cuda(), out, M, N, K, b_kn=True)
    _close(out, a.double() @ b.double())


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (64, 64, 1000), (128, 12, 16384), (512, 512, 8192),
                                   (100, 64, 5000), (129, 257, 777), (512, 512, 131072), (200, 130, 9000), (128, 384, 8200)])
def test_tn_weight_gradient_accumulates(M, N, K):
    from ocean import _hip
    a, b = _rand(K, M, seed=1), _rand(K, N, seed=2)        # reduction over rows
    c0 = _rand(M, N, seed=3)
    out = c0.clone().cuda()
    _hip.gemm(a.cuda(), b.cuda(), out, M, N, K, a_kmajor=True, b_kn=True)
    ref = c0.double() + a.double().t() @ b.double()
    # fp32 accumulation over up to 131072 rows, split-K partial sums
    _close(out, ref, 5e-5)


def test_strided_output_and_unaligned_pointers():
    from ocean import _hip
    M, N, K = 50, 20, 24
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2)
    big = torch.zeros(M, 64, device="cuda")
    view = big[:, 8:]                      # write into columns 8..27 of a wider matrix
    ad, wd = a.cuda(), w.cuda()            # keep the device operands alive across the raw calls
    st = _hip.lib().ocean_gemm_f32(_hip.dev_ptr(ad), K, _hip.dev_ptr(wd), K,
                                   view.data_ptr(), 64, M, N, K, 0, 0, None, None, 0, 0, _hip.stream())
    assert st == 0
    torch.cuda.synchronize()
    _close(big[:, 8:28], a.double() @ w.double().t())
    assert float(big[:, :8].abs().sum()) == 0 and float(big[:, 28:].abs().sum()) == 0
    # operands starting 4 bytes off a 16-byte boundary take the scalar-load path
    ab = torch.zeros(M * K + 1, device="cuda")
    ab[1:] = a.flatten().cuda()
    out = torch.empty(M, N, device="cuda")
    st = _hip.lib().ocean_gemm_f32(ab.data_ptr() + 4, K, _hip.dev_ptr(wd), K, _hip.dev_ptr(out), N,
                                   M, N, K, 0, 0, None, None, 0, 0, _hip.stream())
    assert st == 0
    _close(out, a.double() @ w.double().t())


def test_bad_arguments_are_rejected():
    from ocean import _hip
    t = torch.zeros(4, 4, device="cuda")
    with pytest.raises(_hip.OceanHipError):
        _hip.gemm(t, t, t, 0, 4, 4)
    with pytest.raises(_hip.OceanHipError):
        _hip.gemm(torch.zeros(4, 4), t, t, 4, 4, 4)        # CPU tensor: no fallback


# Shapes that take the k-contiguous-operand kernel (full 128-row tiles, at least 256 of them): plain products ...
@pytest.mark.parametrize("M,N,K", [(32768, 128, 64), (32768, 64, 192), (16384, 512, 48), (32768, 128, 32), (65536, 64, 80)])
def test_large_full_tile_products(M, N, K):
    from ocean import _hip
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    r = _rand(M, N, seed=4)
    out = torch.full((M, N), float("nan"), device="cuda")
    _hip.gemm(a.cuda(), w.cuda(), out, M, N, K, bias=b.cuda(), residual=r.cuda(), relu=True)
    _close(out, torch.relu(a.double() @ w.double().t() + b.double() + r.double()))


# ... and three-tap dilated convolutions whose row tile is one sample (T = 128): the taps read one staged window of
# rows between zero bands; shifts up to 32 rows, either sign order (forward and data-gradient orientation), and a
# dilation beyond the bands or a sample length that does not divide the tile (fall back to the general kernel).
@pytest.mark.parametrize("B,Cin,Cout,d,flip,T", [(256, 64, 128, 1, False, 128), (256, 64, 64, 8, False, 128), (256, 48, 128, 32, True, 128),
                                                 (256, 128, 64, 5, True, 128), (128, 32, 256, 2, False, 128), (256, 64, 128, 40, False, 128),
                                                 (256, 16, 128, 3, False, 128),
                                                 # two and four samples per 128-row tile: a zero band between neighbours
                                                 (512, 64, 128, 4, False, 64), (512, 48, 64, 31, True, 64), (1024, 32, 128, 2, False, 32),
                                                 (1024, 64, 64, 32, True, 32), (512, 64, 128, 3, False, 96)])
def test_dilated_conv_window(B, Cin, Cout, d, flip, T):
    from ocean import _hip
    M = B * T
    x, w = _rand(M, Cin, seed=1), _rand(3, Cout, Cin, seed=2) * 0.1
    xd, wd = x.cuda(), w.cuda()
    out = torch.full((M, Cout), float("nan"), device="cuda")
    s0, st = (d, -d) if flip else (-d, d)
    rc = _hip.lib().ocean_conv_gemm_f32(xd.data_ptr(), Cin, wd.data_ptr(), Cin, out.data_ptr(), Cout, M, Cout, Cin, 0, 0, 3, T, s0, st,
                                        Cout * Cin, 0, 1, _hip.stream())
    assert rc == 0
    torch.cuda.synchronize()
    xs = x.double().view(B, T, Cin)
    ref = torch.zeros(B, T, Cout, dtype=torch.float64)
    for t in range(3):
        s = s0 + t * st
        sh = torch.zeros_like(xs)
        if abs(s) >= T:
            pass
        elif s >= 0:
            sh[:, :T - s] = xs[:, s:]
        else:
            sh[:, -s:] = xs[:, :T + s]
        ref += sh @ w[t].double().t()
    _close(out, ref.view(M, Cout))


@pytest.mark.parametrize("shift", [3, -5])
def test_single_tap_row_shift_at_a_large_aligned_size(shift):
    """One tap with a row shift (out[b, t] = x[b, t + shift] W^T, zero outside the sample) at a size whose tiles are all
    full and fill the chip: the k-contiguous kernel's one-tap body reads rows unshifted, so this problem must be served
    by a kernel that honours the shift."""
    from ocean import _hip
    B, T, Cin, Cout = 256, 128, 64, 128
    M = B * T
    x, w = _rand(M, Cin, seed=1), _rand(Cout, Cin, seed=2) * 0.1
    xd, wd = x.cuda(), w.cuda()
    out = torch.full((M, Cout), float("nan"), device="cuda")
    rc = _hip.lib().ocean_conv_gemm_f32(xd.data_ptr(), Cin, wd.data_ptr(), Cin, out.data_ptr(), Cout, M, Cout, Cin, 0, 0, 1, T, shift, 0,
                                        0, 0, 1, _hip.stream())
    assert rc == 0
    torch.cuda.synchronize()
    xs = x.double().view(B, T, Cin)
    sh = torch.zeros_like(xs)
    if shift >= 0:
        sh[:, :T - shift] = xs[:, shift:]
    else:
        sh[:, -shift:] = xs[:, :T + shift]
    _close(out, (sh @ w.double().t()).view(M, Cout))


def test_window_kernel_random_shapes():
    """Seeded random shapes across the k-contiguous kernel's domain (tile widths 64 / 128, 1 / 2 / 4 samples per tile,
    odd and even numbers of 16-deep chunks, shifts up to the band width, both shift orders) against a float64 reference
    computed on the device."""
    from ocean import _hip
    rng = torch.Generator().manual_seed(1234)
    pick = lambda xs: xs[int(torch.randint(0, len(xs), (1,), generator=rng))]
    for case in range(16):
        T = pick([128, 64, 32])
        Cin, Cout = pick([32, 48, 64, 96, 128, 176]), pick([64, 128, 192, 256])
        rows_tiles = max(256 // (Cout // (128 if Cout % 128 == 0 else 64)), 1)
        M = 128 * (rows_tiles + int(torch.randint(0, 9, (1,), generator=rng)))
        B = M // T
        d = int(torch.randint(1, 33, (1,), generator=rng))
        if d >= T:
            d = T - 1
        flip = bool(torch.randint(0, 2, (1,), generator=rng))
        x = torch.randn(M, Cin, generator=rng).cuda()
        w = (torch.randn(3, Cout, Cin, generator=rng) * 0.1).cuda()
        out = torch.full((M, Cout), float("nan"), device="cuda")
        s0, st = (d, -d) if flip else (-d, d)
        rc = _hip.lib().ocean_conv_gemm_f32(x.data_ptr(), Cin, w.data_ptr(), Cin, out.data_ptr(), Cout, M, Cout, Cin, 0, 0, 3, T, s0, st,
                                            Cout * Cin, 0, 1, _hip.stream())
        assert rc == 0
        xs = x.double().view(B, T, Cin)
        ref = torch.zeros(B, T, Cout, dtype=torch.float64, device="cuda")
        for t in range(3):
            s = s0 + t * st
            sh = torch.zeros_like(xs)
            if s >= 0:
                sh[:, :T - s] = xs[:, s:]
            else:
                sh[:, -s:] = xs[:, :T + s]
            ref += sh @ w[t].double().t()
        err = (out.double() - ref.view(M, Cout)).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0), (case, T, Cin, Cout, M, d, flip, err)


def test_large_kernels_with_padded_leading_dimensions():
    """The k-contiguous product and the long row reduction on operands that are column blocks of wider matrices
    (leading dimensions larger than the logical widths), writing into a column block of a wider output."""
    from ocean import _hip
    L = _hip.lib()
    # NT: A [M, K] inside [M, K + 16], W [N, K] inside [N, K + 32], C [M, N] inside [M, N + 64]
    M, N, K = 32768, 128, 96
    abig, wbig = _rand(M, K + 16, seed=1).cuda(), _rand(N, K + 32, seed=2).cuda()
    cbig = torch.zeros(M, N + 64, device="cuda")
    a, w, c = abig[:, 16:], wbig[:, 32:], cbig[:, 64:]
    assert L.ocean_gemm_f32(a.data_ptr(), K + 16, w.data_ptr(), K + 32, c.data_ptr(), N + 64, M, N, K, 0, 0, None, None, 0, 0, _hip.stream()) == 0
    torch.cuda.synchronize()
    _close(c, (a.double() @ w.double().t()).cpu())
    assert float(cbig[:, :64].abs().sum()) == 0
    # TN: dY [K, M] inside [K, M + 8], X [K, N] inside [K, N + 12], accumulated into dW [M, N] inside [M, N + 4]
    M, N, K = 128, 256, 16384
    ybig, xbig = _rand(K, M + 8, seed=3).cuda(), _rand(K, N + 12, seed=4).cuda()
    dbig = torch.ones(M, N + 4, device="cuda")
    y, x, d = ybig[:, 8:], xbig[:, 12:], dbig[:, 4:]
    assert L.ocean_gemm_f32(y.data_ptr(), M + 8, x.data_ptr(), N + 12, d.data_ptr(), N + 4, M, N, K, 1, 1, None, None, 0, 0, _hip.stream()) == 0
    torch.cuda.synchronize()
    _close(d, (1.0 + y.double().t() @ x.double()).cpu(), 5e-5)
    assert bool((dbig[:, :4] == 1).all())


@pytest.mark.parametrize("M,N,K", [(3072, 256, 256), (1024, 768, 256), (1024, 128, 256), (1024, 12, 64), (1000, 100, 36), (333, 64, 512),
                                   (96, 32, 32), (8192, 64, 64), (20000, 64, 64), (1024, 7, 64), (64, 768, 768), (8, 64, 128), (1, 128, 64), (5, 32, 36)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_linear_layer_gradient_pair(M, N, K, accumulate):
    """dW += dy^T x, db += column sums of dy and dx (+)= dy W of one layer: shapes whose two products share one
    launch (both of few tiles, the reduction of the weight gradient at most 8192 rows), shapes that fall back to two
    launches (N % 4 != 0, M > 8192), ragged tiles, and dx accumulating onto earlier contents."""
    from ocean import _hip
    x, dy, w = _rand(M, K, seed=1), _rand(M, N, seed=2), _rand(N, K, seed=3)
    dw0, db0, dx0 = _rand(N, K, seed=4), _rand(N, seed=5), _rand(M, K, seed=6)
    dw, db, dx = dw0.cuda(), db0.cuda(), dx0.cuda()
    _hip.linear_bwd(x.cuda(), dy.cuda(), w.cuda(), dw, db, dx, M, N, K, accumulate_dx=accumulate)
    _close(dw, dw0.double() + dy.double().t() @ x.double(), tol=4e-5)
    _close(db, db0.double() + dy.double().sum(0), tol=4e-5)
    _close(dx, (dx0.double() if accumulate else 0) + dy.double() @ w.double())
    # no bias gradient, and padded leading dimensions
    xp, dyp, dxp = torch.zeros(M, K + 4), torch.zeros(M, N + 8), torch.full((M, K + 12), 7.0)
    xp[:, :K], dyp[:, :N] = x, dy
    dw, dxd = dw0.cuda(), dxp.cuda()
    _hip.linear_bwd(xp.cuda(), dyp.cuda(), w.cuda(), dw, None, dxd, M, N, K, ldx=K + 4, lddy=N + 8, lddx=K + 12)
    _close(dw, dw0.double() + dy.double().t() @ x.double(), tol=4e-5)
    _close(dxd[:, :K], dy.double() @ w.double())
    assert bool((dxd[:, K:] == 7.0).all())


def test_linear_layer_gradient_pair_refuses_overlap():
    """dx aliasing x: the two products must not share a launch (the data gradient would overwrite the weight gradient's
    operand while it is read); the result must equal weight gradient first, data gradient second."""
    from ocean import _hip
    M, N, K = 1024, 64, 64
    x, dy, w = _rand(M, K, seed=1), _rand(M, N, seed=2), _rand(N, K, seed=3)
    buf, dw = x.cuda(), torch.zeros(N, K, device="cuda")
    _hip.linear_bwd(buf, dy.cuda(), w.cuda(), dw, None, buf, M, N, K)
    _close(dw, dy.double().t() @ x.double(), tol=4e-5)
    _close(buf, dy.double() @ w.double())


def test_small_batch_conv_products_random_shapes():
    """The convolution-shaped products of small batches (fewer than 256 64x64 output blocks: the few-tile kernel's
    variant with taps) across seeded random shapes - 1, 3 or 5 taps, dilations up to beyond the sample length, sample
    lengths that are not multiples of 32, output widths that leave partial tiles, both weight layouts, reductions short
    and long enough for the eight-wave split - against a float64 reference computed on the device."""
    from ocean import _hip
    rng = torch.Generator().manual_seed(4321)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    for case in range(24):
        T = pick([10, 12, 32, 50, 64, 128])
        B = ri(1, max(1, 600 // T))
        M = B * T
        Cin, Cout = 4 * ri(8, 130), ri(5, 300)
        if (M + 63) // 64 * ((Cout + 63) // 64) >= 256:
            Cout = 64
        taps = pick([1, 3, 3, 5])
        d = ri(1, 40)
        flip = bool(ri(0, 1))
        b_kn = bool(ri(0, 1))
        x = torch.randn(M, Cin, generator=rng).cuda()
        w = (torch.randn(taps, Cout, Cin, generator=rng) * 0.1).cuda()            # [tap][n][k]
        wk = w.transpose(1, 2).contiguous() if b_kn else w                          # [tap][k][n]
        out = torch.full((M, Cout), float("nan"), device="cuda")
        half = taps // 2
        s0, st = (half * d, -d) if flip else (-half * d, d)
        rc = _hip.lib().ocean_conv_gemm_f32(x.data_ptr(), Cin, wk.data_ptr(), Cout if b_kn else Cin, out.data_ptr(), Cout, M, Cout, Cin, 0,
                                            int(b_kn), taps, T, s0, st, Cout * Cin, 0, 1, _hip.stream())
        assert rc == 0, (case, rc)
        xs = x.double().view(B, T, Cin)
        ref = torch.zeros(B, T, Cout, dtype=torch.float64, device="cuda")
        for t in range(taps):
            s = s0 + t * st
            sh = torch.zeros_like(xs)
            if abs(s) >= T:
                pass
            elif s >= 0:
                sh[:, :T - s] = xs[:, s:]
            else:
                sh[:, -s:] = xs[:, :T + s]
            ref += sh @ w[t].double().t()
        err = (out.double() - ref.view(M, Cout)).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0), (case, T, B, Cin, Cout, taps, d, flip, b_kn, err)
