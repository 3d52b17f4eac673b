import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_cpu_threads():
    """The CPU oracle is what most of the GPU suite's wall time goes into: on the 256-thread GPU box torch's default thread
    count is its slowest setting by far (oneDNN / OpenMP oversubscription: 13.7 s per 1080p step on 128 threads, > 150 s on
    256, 4.7 s on 16 -- profiles/r02_cpu_threads.txt).  Results do not depend on it to the tested tolerances."""
    import torch
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, avail)))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_backend():
    from tests.backends import OracleBackend
    return OracleBackend()


@pytest.fixture(scope="session")
def hip_backend():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.backends import HipBackend
    return HipBackend()
