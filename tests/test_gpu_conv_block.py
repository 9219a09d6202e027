"""DilatedConvBlock called on its own (the reference exports it; forward: dilated_cnn.py:76-99) through the HIP path
against the same layers in plain torch on the CPU.  fp32; outputs 1e-5 (1e-4 through BatchNorm at few rows), gradients
1e-3, running statistics 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def close(got, ref, tol, what=""):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    scale = max(ref.abs().max().item(), 1e-3) if ref.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} scale {scale:.3e}"


class TorchBlock(torch.nn.Module):
    """dilated_cnn.py:21-99 with torch layers (dropout left out: the tests run it at p = 0 or in eval mode)."""

    def __init__(self, cin, cout, k, d):
        super().__init__()
        self.conv = torch.nn.Conv1d(cin, cout, k, dilation=d, padding=(k - 1) * d // 2, bias=False)
        self.batch_norm = torch.nn.BatchNorm1d(cout)
        self.use_residual = cin == cout
        if not self.use_residual:
            self.residual_conv = torch.nn.Conv1d(cin, cout, 1, bias=False)
            self.residual_bn = torch.nn.BatchNorm1d(cout)

    def forward(self, x):
        out = torch.relu(self.batch_norm(self.conv(x)))
        res = x if self.use_residual else self.residual_bn(self.residual_conv(x))
        return out + res


@pytest.mark.parametrize("B,T,cin,cout,k,d", [(2, 16, 8, 8, 3, 1), (3, 33, 12, 20, 3, 2), (4, 64, 64, 128, 3, 4), (2, 10, 6, 6, 5, 2),
                                              (5, 7, 16, 24, 1, 1), (64, 128, 64, 64, 3, 8)])
@pytest.mark.parametrize("training", [True, False])
def test_dilated_conv_block_matches_torch_layers(B, T, cin, cout, k, d, training):
    from ocean.models.components import DilatedConvBlock
    torch.manual_seed(B + T + cin)
    m = DilatedConvBlock(cin, cout, k, d, dropout=0.0 if training else 0.1)
    with torch.no_grad():
        for bn in [m.batch_norm] + ([] if m.use_residual else [m.residual_bn]):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    ref = TorchBlock(cin, cout, k, d)
    ref.load_state_dict(m.state_dict())
    m, _ = m.cuda().train(training), ref.train(training)
    x, w = torch.randn(B, cin, T), torch.randn(B, cout, T)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    assert y.shape == (B, cout, T) and y.is_contiguous()
    close(y, yr, 1e-4, "output")
    (y * w.cuda()).sum().backward()
    close(xg.grad, xr.grad, 1e-3, "dx")
    for name, p in m.named_parameters():
        close(p.grad, dict(ref.named_parameters())[name].grad, 1e-3, name)
    for name, b in m.named_buffers():
        close(b, dict(ref.named_buffers())[name], 1e-5, name)


def test_dilated_conv_block_dropout_and_errors():
    from ocean.models.components import DilatedConvBlock
    torch.manual_seed(0)
    m = DilatedConvBlock(16, 16, 3, 2, dropout=0.3).cuda().train()
    x = torch.randn(8, 16, 32).cuda().requires_grad_(True)
    y = m(x)
    main = y - x                                     # Dropout(ReLU(BatchNorm(conv))): zero where dropped or gated
    frac = (main.abs() < 1e-7).float().mean().item()
    assert 0.3 < frac < 0.9, frac
    y.square().mean().backward()
    assert bool(torch.isfinite(x.grad).all()) and m.conv.weight.grad is not None
    assert not torch.equal(m(x.detach()), y)         # another mask
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m(torch.randn(1, 16, 1).cuda())
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 8, 32).cuda())


def test_dilated_conv_block_random_shapes():
    """Seeded random widths (odd ones too), lengths shorter and longer than the receptive field, kernel sizes 1-7,
    dilations up to 16, both residual forms, training mode (batch statistics, running-statistics update)."""
    import random
    from ocean.models.components import DilatedConvBlock
    rnd = random.Random(11)
    for case in range(16):
        cin = rnd.choice([1, 3, 8, 17, 32, 64])
        cout = cin if rnd.random() < 0.5 else rnd.choice([2, 5, 16, 48, 128])
        k, d = rnd.choice([1, 3, 5, 7]), rnd.choice([1, 2, 3, 8, 16])
        B, T = rnd.choice([1, 2, 7]), rnd.choice([2, 3, 16, 40, 128])
        torch.manual_seed(case)
        m = DilatedConvBlock(cin, cout, k, d, dropout=0.0)
        ref = TorchBlock(cin, cout, k, d)
        ref.load_state_dict(m.state_dict())
        m = m.cuda().train()
        x, w = torch.randn(B, cin, T), torch.randn(B, cout, T)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        (yr * w).sum().backward()
        xg = x.cuda().requires_grad_(True)
        y = m(xg)
        what = f"case {case}: B={B} T={T} {cin}->{cout} k={k} d={d}"
        close(y, yr, 1e-4, what + " output")
        (y * w.cuda()).sum().backward()
        # few rows per channel: BatchNorm's backward is a difference of nearly equal terms there
        gtol = 1e-2 if B * T <= 4 else 1e-3
        close(xg.grad, xr.grad, gtol, what + " dx")
        for name, p in m.named_parameters():
            close(p.grad, dict(ref.named_parameters())[name].grad, gtol, what + " " + name)
        for name, b in m.named_buffers():
            close(b, dict(ref.named_buffers())[name], 1e-5, what + " " + name)
