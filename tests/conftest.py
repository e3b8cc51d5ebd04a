import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 while librlhip.so uses /opt/rocm's.  Both
    # runtimes coexist in one process only if torch's is initialised FIRST (the other order leaves torch with "No HIP
    # GPUs are available"); bench.py has that order by construction, tests that mix the two get it here.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return has_gpu()
