import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """every GPU test gets a 10-minute ceiling (pytest-timeout) unless it sets its own: a wedged worker process or collective must
    fail the test, not hang the run"""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def gold():
    return golden


@pytest.fixture(autouse=True)
def _fresh_x6_range_flags(request):
    """GPU tests: clear the bf16x6 range-guard word before every test -- it is process-wide device state (abr_x6_range_flags), and a test that
    legitimately raises it (tiny operands of a padded batch, injected inf / nan) must not leak into the next test's `== 0` assertion."""
    if request.node.get_closest_marker("gpu") is not None:
        try:
            import torch
            if torch.cuda.is_available():
                from abr_iod_amd import ops
                ops.x6_range_flags(reset=True)
        except Exception:
            pass
    yield
