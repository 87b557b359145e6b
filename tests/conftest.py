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
    """plain `pytest tests` on a box without a GPU: skip the gpu-marked tests instead of failing them.  An explicit
    `-m gpu` run is never skipped: there a missing GPU / library must fail loudly."""
    if "gpu" in (config.getoption("markexpr", "") or ""):
        return
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:       # noqa: BLE001
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def ora():
    from oracle import ssg_oracle
    ssg_oracle.build()
    return ssg_oracle


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint16) if a.dtype == np.float16 else a


sys.path.insert(0, os.path.join(ROOT, "tools"))
from synth import clustered, hard_clustered  # noqa: E402,F401  (Track-G synthetic embeddings; same generator as tools/make_golden.py)
