import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A -m gpu run on a box without a GPU must fail loudly, not skip silently."""
    import torch
    if torch.cuda.is_available():
        return
    for item in items:
        if "gpu" in item.keywords and config.getoption("-m") and "not gpu" not in config.getoption("-m"):
            item.add_marker(pytest.mark.xfail(reason="no GPU visible", run=False, strict=True))
