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


def thresholds_until_round_6(engine):
    """The mixed paths' minimum launch sizes as they were until round 6 (first generation from 2^17 pairs, second from 2^25,
    third from 2^18).  Round 6 moved them (2^20 / 2^20 / 3 * 2^20: below, the direct path's cell-table kernel is faster, and
    the first generation is no device-resident call's default any more); the tests written against the old ones keep them, so
    that the first generation and the small launches of the others stay under test."""
    from loghisto_amd import _native as N
    engine.set_option(N.OPT_PART_MIN_PAIRS, 1 << 17)
    engine.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 25)
    engine.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 18)
