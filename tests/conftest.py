import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): test infrastructure only."""
    from oracle import pointnet2_ref
    pointnet2_ref.build()
    return pointnet2_ref


@pytest.fixture(scope="session")
def hip_ext():
    """pointnet2._ext over libsig3d_hip.so; fails (not skips) if the library is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    import pointnet2._ext as ext
    return ext
