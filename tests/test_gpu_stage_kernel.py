"""The opt-in stage kernels (OCEAN_STAGE_FUSE=1: the attention / fusion / head chains of a small batch as one kernel
each, csrc/rowchain.hip) against the default per-layer launches on the same seeded step: same operations in the same
order per row, so probabilities and the loss agree to rounding and every gradient to 1e-4 of its scale.  Batches of 40
and 128 rows put several 16-sample blocks on the kernel, the last one partly filled at 40."""
import os
import subprocess
import sys

import pytest
import torch

from stage_kernel_step import run_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,dropout", [(40, 0.0), (128, 0.0), (40, 0.1)])
def test_stage_kernels_reproduce_the_per_layer_launches(tmp_path, B, dropout):
    assert os.environ.get("OCEAN_STAGE_FUSE", "0") in ("", "0"), "this process must run the default launches"
    ref = run_step(B, dropout)
    out = tmp_path / "fused.pt"
    env = dict(os.environ, OCEAN_STAGE_FUSE="1")
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stage_kernel_step.py")
    done = subprocess.run([sys.executable, child, str(out), str(B), str(dropout)], env=env, capture_output=True, text=True, timeout=180)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "HSA_STATUS_ERROR" not in done.stderr, done.stderr[-2000:]
    got = torch.load(out, weights_only=True)
    assert set(got) == set(ref)
    for k in ref:
        a, b = got[k].double(), ref[k].double()
        tol = 1e-5 if k in ("probs", "total") else 1e-4
        err, scale = (a - b).abs().max().item(), max(b.abs().max().item(), 1e-6)
        assert err <= tol * scale, (k, err, scale)
