import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "unsupervised-pseuso-lidar_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny) with numpy arrays or torch tensors."""
    import torch
    a = torch.as_tensor(np.asarray(a.detach().cpu() if hasattr(a, "detach") else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if hasattr(b, "detach") else b), dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
