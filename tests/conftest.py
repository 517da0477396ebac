import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Build (or reuse) libvse_hip.so; cross-compiles for gfx950 without a GPU."""
    import __graft_entry__
    return __graft_entry__.build()


@pytest.fixture(scope="session")
def ctx(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible: the product path has no CPU fallback")
    from vse_amd import engine
    return engine.Context(0)
