import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The oracle's sweeps are OpenMP loops over rows, one fork/join per time step, and most CPU tests
# step small lattices thousands of times.  On a shared host whose cores are taken away now and then,
# every join waits for the slowest of the threads; with all cores in the team the same test was
# seen to take 4 s or 70 s.  Half the cores unless the environment says otherwise (set before the
# OpenMP runtime loads).
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(4, (os.cpu_count() or 2) // 2))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The library, ./main and bench.py default to the contracted arithmetic.  The suite's reference
# point is the strict one (bit-identical to the oracle's literal restatement of pull-push.cpp); tests
# of the other arithmetics select them explicitly.  Set before the library is loaded, inherited by
# every child process (command-line driver, ranks of the multi-process tests).
os.environ.setdefault("LBM_ARITHMETIC", "strict")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no HIP device is visible."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def ranks_share_one_gpu():
    """For tests that start several ranks on cuda:0 (RCCL refuses two ranks on one GPU, so these are
    rehearsals over gloo).  Ranks that find another rank of their group on their device take their
    edge stream at default priority (lbm_amd.slab.ranks_share_device): a process's high-priority
    hardware queues were what stalled when it took cross-queue event waits, default-priority
    streams never did.  The children run with the HIP runtime's own queue limit; every rank arms a
    watchdog that prints its stacks and exits, so a stall would fail one test with its location
    instead of hanging the suite."""
    yield
