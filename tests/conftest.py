import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# kernel-variant switches of the library (SKIMI_GEMM256_*) are re-read on every launch, so that one
# test process can walk through the variants (gemm256.hip: env_int)
os.environ.setdefault("SKIMI_ENV_DYNAMIC", "1")
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; on a machine without a device they are skipped
    # (never silently passed through a fallback).
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
