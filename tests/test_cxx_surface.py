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
