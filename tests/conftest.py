import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cap_cpu_threads():
    """The oracle (PyTorch CPU) does most of this suite's CPU work.  On a GPU box the process sees every core of the host
    while its share is a fraction of them (16 for one GPU): torch's default of one thread per visible core oversubscribes
    that share several times over (the driver's box ran the GPU suite in 498 s, the builder's in 104 s).  Cap the pools."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, 16))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):  # (inherited by the child interpreters of the multi-process tests)
        os.environ.setdefault(var, str(n))
    try:
        import torch
        torch.set_num_threads(n)
    except Exception:
        pass


def pytest_configure(config):
    _cap_cpu_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the longer GPU sweeps (still part of -m gpu; deselect with -m 'gpu and not slow')")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
