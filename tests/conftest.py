import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip_library():
    """The in-tree HIP library; GPU tests fail loudly when it is not built."""
    from joshupscale_amd import runtime
    return runtime.load_library()


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first(request):
    """torch ships its own HIP runtime; when it initialises AFTER libJoshUpscale.so
    has opened the device in the same process it can report "No HIP GPUs are
    available".  bench.py initialises torch first; do the same for GPU test runs."""
    markexpr = request.config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda").cpu()
    yield
