import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _arithmetic_mode_does_not_leak(request):
    """A GPU test that switches the library's arithmetic mode (ssv_set_precision) must not change it for the tests that run after it in
    the same process: whatever the test did, the mode it found is put back."""
    if "gpu" not in request.keywords:
        yield
        return
    from spoofsv_amd import _lib
    before = _lib.precision()
    yield
    if _lib.precision() != before:
        _lib.lib().ssv_set_precision(before)
