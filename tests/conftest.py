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


def probed_introsort_threshold():
    """insertion-sort threshold of the INSTALLED numpy's argsort on half keys (15 or 16: `pr - pl > t` is still partitioned), probed with
    17-element tie patterns on which the oracle's two settings order differently; None when neither matches"""
    from oracle import ssg_oracle as ora
    ora.build()
    rng = np.random.default_rng(11)
    votes = {15: 0, 16: 0}
    for _ in range(600):
        h = (rng.integers(0, 4, 17) / 4.0).astype(np.float16)
        a15, a16 = ora.argsort_half(h, small=15), ora.argsort_half(h, small=16)
        if not np.array_equal(a15, a16):
            got = np.argsort(h)
            votes[15] += int(np.array_equal(got, a15)); votes[16] += int(np.array_equal(got, a16))
    if votes[15] and not votes[16]:
        return 15
    if votes[16] and not votes[15]:
        return 16
    return None


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """one line in every log (also with -q, also in the -m gpu log of the GPU box): the numpy build whose introsort tie order the
    default rank mode replays, and the threshold probed on it -- "bit-identical to the unmodified reference" is a statement about
    THIS numpy"""
    try:
        thr = probed_introsort_threshold()
    except Exception as e:       # noqa: BLE001  (the oracle could not be built: say so, do not fail the run)
        thr = "probe failed: %s" % (e,)
    terminalreporter.write_line("ssg: numpy %s, probed introsort insertion threshold %s (kernel / oracle / goldens: 15); python %s"
                                % (np.__version__, thr, sys.version.split()[0]))


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
