import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Builds (if stale) and loads liblhgpu.so.  hipcc cross-compiles without a GPU."""
    from loghisto_amd import _native, build
    build.build_native()
    return _native.lib()


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch
