import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The test session works through libJoshUpscale_test.so: the product library does not export the hooks the
# tests use (ju_debug_*, ju_read_tensor, ju_time_steps: include/joshupscale_amd_test.h).  The product library
# itself is held to its symbol list (tests/test_c_abi.py) and to the test flavour's bytes
# (tests/test_gpu_parity.py::test_product_library_gives_the_test_flavours_bytes).
os.environ.setdefault("JU_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip_library():
    """The in-tree HIP library.  The product never builds or falls back by itself
    (runtime.load_library raises when the .so is missing); the TEST session builds it
    once when it is absent and hipcc is available, so a fresh checkout can run the CPU
    suite (hipcc cross-compiles gfx950 without a GPU)."""
    import shutil
    import subprocess
    from joshupscale_amd import runtime
    if "JU_LIBRARY" not in os.environ and not (os.path.exists(runtime.library_path(True)) and
                                               os.path.exists(runtime.library_path(False))):
        if shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.check_call(["make", "-s", "-j8", "-C", ROOT])
    return runtime.load_library(True)


@pytest.fixture(scope="session")
def product_library(hip_library):
    """The library a plugin host loads: no hooks."""
    from joshupscale_amd import runtime
    return runtime.load_library(False)


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


def pytest_sessionfinish(session, exitstatus):
    """GPU runs leave the measured parity numbers behind (gpurun_out/parity_stats.json)."""
    try:
        import gpu_common
        gpu_common.dump_stats()
    except Exception:  # pragma: no cover - never fail a run over the report
        pass
