import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ocean-online-multi-modal-root-cause-analysis_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _dropout_epoch_starts_at_zero(request):
    """The dropout masks of the HIP path depend on a per-device epoch counter that recorded update steps advance.
    Tests that fix the mask seeds (finite differences through dropout) must not depend on how many replays earlier
    tests ran: every GPU test starts from epoch 0."""
    if request.node.get_closest_marker("gpu") is not None:
        from ocean import _hip
        for t in _hip._epochs.values():
            t.zero_()
    yield
