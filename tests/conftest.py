import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU work (compiles the whole library to assembly); deselect with -m 'not slow'")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "ref_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ops_ref
    ops_ref.build()
    return ops_ref


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionstart(session):
    """The oracle is torch-CPU + OpenMP: cap its threads (a 256-thread run of these small problems on the GPU host is far
    slower than 32 threads)."""
    n = min(os.cpu_count() or 1, 32)
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    try:
        import torch
        torch.set_num_threads(n)
    except Exception:
        pass
