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


@pytest.fixture(scope="session", autouse=True)
def _bandwidth_hog():
    """UNFLOW_TEST_HOG=1: run the GPU suite beside a second PROCESS that keeps the GPU's memory system busy with 600 MB streaming
    copies — every kernel's timing changes (memory latency, L2 / Infinity-Cache contents), its results must not (round 6: a register
    hazard in the 81-channel correlation kernel was invisible to the quiet suite).  A process, not a thread: a host thread that
    launches or synchronises while another one captures a hipGraph invalidates the capture."""
    if os.environ.get("UNFLOW_TEST_HOG") != "1":
        yield
        return
    import subprocess
    code = ("import torch, time\n"
            "d = torch.device('cuda:0'); n = 150_000_000\n"
            "a, b = torch.randn(n, device=d), torch.empty(n, device=d)\n"
            "t0 = time.time()\n"
            "while time.time() - t0 < 3000:\n"
            "    for _ in range(8): b.copy_(a)\n"
            "    torch.cuda.synchronize()\n")
    p = subprocess.Popen([sys.executable, "-c", code])
    import time
    time.sleep(8)          # let it start streaming
    yield
    p.kill()
    p.wait()
