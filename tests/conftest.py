"""pytest configuration: markers, import path, shared fixtures."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a machine without a GPU must fail loudly, not skip: the driver
    # records silently-skipped GPU tests as "native code not loaded".
    # `-m "not gpu"` simply deselects them.  With no -m at all, skip GPU tests on
    # CPU-only machines so a plain `pytest tests/` is still usable here.
    if config.getoption("-m"):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name: str):
        return np.load(GOLDEN / name, allow_pickle=False)

    return load
