import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """``-m gpu`` tests call the HIP library: skip them (instead of failing at the first assert) on a box
    without a GPU, e.g. a plain ``pytest tests`` in the build container."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
