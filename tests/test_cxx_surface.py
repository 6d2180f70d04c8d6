"""The C++ plugin surface (include/JoshUpscale/core.h) from a real C++ caller, on CPU:
member names of Runtime and the exact getExceptionString() format of the reference
(core/public/JoshUpscale/core.h:64-94, core/src/exception.cc:51-79)."""

import os
import re
import subprocess

from helpers import ROOT
from joshupscale_amd import runtime as R


def test_exception_string_format_and_member_names(hip_library, tmp_path):
    exe = str(tmp_path / "exception_format")
    lib_dir = os.path.dirname(R.library_path())
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "exception_format.cpp"), "-o", exe,
                           "-L" + lib_dir, "-lJoshUpscale", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    blocks = re.findall(r"\[(.*?)\]\n", out.stdout, flags=re.S)
    assert len(blocks) == 3, out.stdout
    # nested: "\n  " before EVERY nested exception (not a growing indent), no trailing newline
    assert re.fullmatch(r"std::_Nested_exception<std::logic_error>: outer\n"
                        r"  std::_Nested_exception<std::runtime_error>: middle\n"
                        r"  std::invalid_argument: innermost", blocks[0]), blocks[0]
    assert blocks[1] == "Unknown error"
    assert blocks[2].startswith("std::ios_base::failure") and "cannot open model file" in blocks[2]
    assert not blocks[2].endswith("\n")


def test_reference_plugin_call_sites_compile_against_the_header():
    """Every use the reference's AviSynth and OBS plugins make of JoshUpscale::core
    (avisynth_plugin/src/main.cc:40, 57, 62-68, 113-148; obs_plugin/src/filter.cc:64-69, 247-275,
    291-306, 384-389; plugin.cc:93-106; logging.cc:20-48), restated in tests/cxx/plugin_call_sites.cpp,
    compiles against include/JoshUpscale/core.h with both compilers of the image, warnings as errors.
    The reference builds with C++20 (designated initialisers of core::Image)."""
    src = os.path.join(ROOT, "tests", "cxx", "plugin_call_sites.cpp")
    compilers = ["g++"]
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if os.path.exists(clang):
        compilers.append(clang)
    for cxx in compilers:
        out = subprocess.run([cxx, "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                              "-I" + os.path.join(ROOT, "include"), src], capture_output=True, text=True)
        assert out.returncode == 0, (cxx, out.stderr)
