"""The hand-counted `s_waitcnt lgkmcnt(N)` of the built kernels, recounted on the machine code (CPU: disassembly only).

csrc/tower_kernels.hip, tower8_kernels.hip, fp8_kernels.hip, flow_kernels.hip ... issue their LDS fragment reads as
inline asm and wait for exactly the reads an MFMA needs; the count in the source includes LDS traffic the COMPILER
emits between them ("if a future compiler merges them (`ds_read2`), splits them, or sinks them, the wait
becomes too lenient and MFMAs read stale fragments without any error").  tools/lds_wait_check.py walks the disassembly of
the library that ships with the queue of outstanding LDS operations and reports every instruction that touches a
register whose read a wait has not yet covered -- on every build, whatever the toolchain."""

import importlib.util
import os

import pytest

from helpers import ROOT

LIB = os.path.join(ROOT, "joshupscale_amd", "lib", "libJoshUpscale.so")


def _tool():
    spec = importlib.util.spec_from_file_location("lds_wait_check", os.path.join(ROOT, "tools", "lds_wait_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _lines(text):
    return [(f"{i:04x}", *(ln.split(None, 1) + [""])[:2]) for i, ln in enumerate(text.strip().splitlines())]


def test_the_checker_catches_a_wait_that_is_too_lenient():
    t = _tool()
    ok = """
ds_read_b128 v[10:13], v1
ds_read_b128 v[14:17], v1 offset:4352
ds_read_b128 v[18:21], v1 offset:8704
s_waitcnt lgkmcnt(2)
v_mfma_f32_32x32x16_bf16 v[32:47], a[0:3], v[10:13], v[32:47]
s_waitcnt lgkmcnt(1)
v_mfma_f32_32x32x16_bf16 v[32:47], a[4:7], v[14:17], v[32:47]
ds_write_b64 v2, v[50:51]
s_waitcnt lgkmcnt(1)
v_mfma_f32_32x32x16_bf16 v[32:47], a[8:11], v[18:21], v[32:47]
"""
    v, stats = t.check_kernel(_lines(ok))
    assert v == [] and stats["counted_waits"] == 3 and stats["ds"] == 4
    # the source counted TWO compiler loads between the fragment read and its use; the compiler merged them into one
    merged = """
ds_read_b128 v[10:13], v1
ds_read2_b64 v[60:63], v3 offset1:8
s_waitcnt lgkmcnt(2)
v_mfma_f32_32x32x16_bf16 v[32:47], a[0:3], v[10:13], v[32:47]
"""
    v, _ = t.check_kernel(_lines(merged))
    assert len(v) == 1 and ("v", 10) in v[0][2]
    # a register reused while a (dead) read into it is still in flight
    reused = """
ds_read_b128 v[54:57], v245
buffer_load_dwordx4 v[54:57], v212, s[68:71], 0 offen sc1
"""
    v, _ = t.check_kernel(_lines(reused))
    assert len(v) == 1
    # the two sides of an if / else on EXEC run on disjoint lanes (registers are per lane): the else side may reuse
    # a register whose read is in flight on the then side -- but not one whose read is in flight on its OWN side,
    # and after the join every read of both sides counts again
    if_else = """
s_and_b64 s[4:5], s[2:3], s[6:7]
s_xor_b64 s[34:35], s[4:5], s[2:3]
s_mov_b64 exec, s[4:5]
s_cbranch_execz 10
ds_read_b128 v[86:89], v74 offset:8704
ds_read_b128 v[74:77], v74 offset:13056
s_andn2_saveexec_b64 s[34:35], s[34:35]
s_cbranch_execz 21
v_accvgpr_read_b32 v86, a133
ds_read_b128 v[94:97], v86
ds_read_b128 v[86:89], v86 offset:8704
s_or_b64 exec, exec, s[34:35]
s_waitcnt lgkmcnt(0)
v_mfma_f32_32x32x16_bf16 v[32:47], a[0:3], v[86:89], v[32:47]
"""
    v, stats = t.check_kernel(_lines(if_else))
    assert v == [] and stats["ds"] == 4
    own_side = if_else.replace("v_accvgpr_read_b32 v86, a133\nds_read_b128 v[94:97], v86",
                               "v_accvgpr_read_b32 v86, a133\nds_read_b128 v[94:97], v86\nv_mov_b32_e32 v94, 0")
    v, _ = t.check_kernel(_lines(own_side))
    assert len(v) == 1 and ("v", 94) in v[0][2]
    after_join = if_else.replace("s_waitcnt lgkmcnt(0)\n", "")
    v, _ = t.check_kernel(_lines(after_join))
    assert len(v) == 1 and ("v", 86) in v[0][2]
    # scalar memory returns out of order: a counted wait proves nothing while one is outstanding
    smem = """
s_load_dwordx2 s[4:5], s[0:1], 0x0
ds_read_b128 v[10:13], v1
ds_read_b128 v[14:17], v1 offset:4352
s_waitcnt lgkmcnt(1)
v_mfma_f32_32x32x16_bf16 v[32:47], a[0:3], v[10:13], v[32:47]
"""
    v, stats = t.check_kernel(_lines(smem))
    assert len(v) == 1 and stats["out_of_order_under_counted_wait"] == 1


def _family(report, fam):
    """The statistics of every instantiation of kernel template `fam`: by the Itanium-ABI spelling of its NAME
    (<length><identifier>I<template arguments>), not by fragments of its arguments -- a toolchain or
    template-parameter change must not fail this test for reasons that have nothing to do with waits."""
    return [stats for k, (_, stats) in report.items() if f"{len(fam)}{fam}I" in k]


def test_every_counted_lds_wait_of_the_shipped_library_covers_its_reads():
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(LIB):
        pytest.skip(f"NOT CHECKED: {LIB} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not (os.path.exists(objdump) or shutil.which("llvm-objdump")):
        pytest.skip("NOT CHECKED: no llvm-objdump on this machine, the shipped kernels' waits were not recounted")
    t = _tool()
    report = t.check_library(LIB)
    bad = {k: v[0][:3] for k, v in report.items() if v[0]}
    assert not bad, bad
    # ... and the kernels the check exists for were really walked.  By kernel NAME, not by template-argument fragments or
    # fixed counts: every hand-scheduled kernel family
    # is present, and each of them carries counted waits.
    families = ["tower_resident_kernel", "tower8_resident_kernel", "flow_block_kernel", "res_block_pipe_kernel",
                "res_block_fp8_kernel", "conv_splitk_kernel", "conv_tower_kernel"]
    for fam in families:
        mine = _family(report, fam)
        assert mine, f"no {fam} instantiation in {LIB}"
        assert sum(s["counted_waits"] for s in mine) > 0, fam
    # the resident tower's instantiations are the densest users: hundreds of counted waits each
    tower = _family(report, "tower_resident_kernel")
    assert max(s["counted_waits"] for s in tower) > 500, [s["counted_waits"] for s in tower]
    assert len(report) >= len(families)
