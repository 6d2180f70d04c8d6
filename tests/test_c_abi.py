"""The C-ABI library loads and exports every symbol include/*.h declares (CPU).
No compute call is made here: there is no GPU in the CPU test environment."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import M, ROOT
from joshupscale_amd import runtime as R


def declared_functions(header="joshupscale_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"JU_API\s+[\w\s\*]+?\b(ju_\w+)\s*\(", text)))


def exported(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return [line.split()[-1] for line in out.splitlines() if line.strip()]


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["ju_create", "ju_create_from_memory", "ju_destroy", "ju_process", "ju_get_size",
                 "ju_reset", "ju_last_error", "ju_set_log_callback"]:
        assert must in names


def test_library_exports_every_declared_symbol(hip_library, product_library):
    for name in declared_functions():
        assert hasattr(product_library, name), name
        assert hasattr(hip_library, name), name
    for name in declared_functions("joshupscale_amd_test.h"):
        assert hasattr(hip_library, name), name


def test_product_library_exports_exactly_the_declared_surface(product_library):
    """The shipped library exports the C ABI of include/joshupscale_amd.h and the C++ plugin surface of
    include/JoshUpscale/core.h -- and no test or developer hook (the reference hides everything that is not
    JOSHUPSCALE_EXPORT: core/CMakeLists.txt:29-36)."""
    names = exported(R.library_path(False))
    c_abi = sorted(n for n in names if n.startswith("ju_"))
    assert c_abi == declared_functions(), set(c_abi) ^ set(declared_functions())
    assert tuple(c_abi) == R.PRODUCT_SYMBOLS
    hooks = declared_functions("joshupscale_amd_test.h")
    assert sorted(hooks) == list(R.HOOK_SYMBOLS)
    assert not [n for n in names if "debug" in n.lower() or n in hooks]
    # the C++ surface: the five non-Windows functions of core.h:28, 60-62, 91-94 (plus weak template / typeinfo
    # symbols of the standard library, which every C++ shared object carries)
    strong = subprocess_nm_strong(R.library_path(False))
    cxx = sorted(n for n in strong if n.startswith("_ZN11JoshUpscale"))
    assert len(cxx) == 5 and all(re.search(r"core\d+(createRuntime|getExceptionString|setLogSink|getGLImage|getGLDeviceIndex)", n)
                                 for n in cxx), cxx
    other = [n for n in strong if not n.startswith(("ju_", "_ZN11JoshUpscale", "__hip_cuid_"))]
    assert not other, other
    # the test flavour: the same surface + exactly the hooks
    tnames = exported(R.library_path(True))
    assert sorted(n for n in tnames if n.startswith("ju_")) == sorted(declared_functions() + hooks)


def test_product_library_reads_only_the_documented_environment(product_library):
    """Every developer switch (A/B paths, cross-check kernels, traces) goes through csrc/dev_switch.h and exists in the
    test flavour only: the product library holds the NAMES of exactly the variables INTEGRATION.md documents for it."""
    import subprocess

    def ju_strings(path):
        out = subprocess.check_output(["strings", "-n", "4", path]).decode(errors="replace")
        return sorted({m.group(0) for line in out.splitlines() for m in [re.match(r"JU_[A-Z0-9_]+", line)] if m})

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("### The product library (`libJoshUpscale.so`)"):text.index("### The test flavour")]
    documented = sorted(set(re.findall(r"^\| `(JU_[A-Z0-9_]+)", section, flags=re.M)))
    assert documented == ["JU_LOOKAHEAD", "JU_NO_GRAPH", "JU_RESIDENT_RETRY", "JU_VERBOSE"]
    assert ju_strings(R.library_path(False)) == documented
    # the test flavour does carry the developer switches (the same objects + dev_switch.cpp under -DJU_TEST_HOOKS)
    dev = ju_strings(R.library_path(True))
    assert set(documented) < set(dev) and {"JU_TOWER", "JU_FLOW_CONV", "JU_TAIL", "JU_FP8_GRID"} <= set(dev)
    # ... and every one of them is documented in the second table
    tail = text[text.index("### The test flavour"):]
    listed = set(re.findall(r"`(JU_[A-Z0-9_]+)", tail))
    assert set(dev) - set(documented) <= listed, set(dev) - set(documented) - listed


def subprocess_nm_strong(path):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return [line.split()[-1] for line in out.splitlines() if line.split()[1] in ("T", "D", "B", "R")]


def test_cxx_plugin_surface_is_exported():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", R.library_path(False)]).decode()
    for sym in ["createRuntime", "getExceptionString", "setLogSink", "getGLImage",
                "getGLDeviceIndex"]:
        assert re.search(r"_ZN11JoshUpscale4core\d+" + sym, out), sym


def test_image_struct_matches_core_h_layout():
    # struct Image {void*; uint8; ptrdiff_t; size_t; size_t} (reference core.h:32-38)
    assert C.sizeof(R.JuImage) == 40
    assert R.JuImage.stride.offset == 16 and R.JuImage.width.offset == 24


def test_version_and_error_paths_without_gpu(hip_library, tmp_path):
    assert hip_library.ju_version().decode().startswith("joshupscale-amd")
    h = C.c_void_p()
    rc = hip_library.ju_create(0, str(tmp_path / "missing.jupw").encode(), C.byref(h))
    assert rc == 2 and not h.value                      # JU_ERR_IO
    assert b"cannot open model file" in hip_library.ju_last_error()
    assert hip_library.ju_process(None, None, None) == 1  # JU_ERR_INVALID_ARGUMENT
    assert hip_library.ju_get_gl_device_index(None) == 1  # JU_ERR_INVALID_ARGUMENT
    dev = C.c_int(7)
    assert hip_library.ju_get_gl_device_index(C.byref(dev)) in (3, 5) and dev.value == -1   # no GPU / no GL context here
    img = R.JuImage()
    assert hip_library.ju_get_gl_image(1, 0, C.byref(img)) != 0 and not img.ptr          # no current GL context
    assert hip_library.ju_get_gl_image(1, 9, C.byref(img)) == 1                           # bad image type
    hip_library.ju_release_gl_image(C.byref(img))                                          # no-op on an empty image
    hip_library.ju_destroy(None)                          # no-op


def test_log_callback_receives_errors(hip_library):
    seen = []
    cb = R.LOG_CALLBACK(lambda tag, lvl, msg, user: seen.append((tag, lvl, msg)))
    hip_library.ju_set_log_callback(cb, None)
    try:
        hip_library.ju_reset(None)
    finally:
        hip_library.ju_set_log_callback(R.LOG_CALLBACK(0), None)
    assert seen and seen[0][1] == 2 and b"runtime is NULL" in seen[0][2]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.delenv("JU_LIBRARY", raising=False)
    monkeypatch.setattr(R, "_LIB_NAME", "libDoesNotExist.so")
    monkeypatch.setattr(R, "_TEST_LIB_NAME", "libDoesNotExist_test.so")
    for hooks in (False, True):
        with pytest.raises(ImportError, match="no CPU fallback"):
            R.load_library(hooks)


def test_hooks_are_refused_by_the_product_library(product_library):
    with pytest.raises(RuntimeError, match="test hook"):
        R._hook(product_library, "ju_read_tensor")
