#!/usr/bin/env python3
"""Static check of the hand-counted `s_waitcnt lgkmcnt(N)` in the built kernels.

The hot loops of `tower_resident_kernel`, `conv_tower_kernel`, `tower8_resident_kernel`, ... issue their LDS
fragment reads as inline asm and wait with a COUNTED `s_waitcnt lgkmcnt(N)`: "all but the N youngest LDS
operations have returned" (LDS operations return in order).  N is computed in the source from what the source
believes lies between a read and its use -- including LDS traffic the COMPILER emits (residual loads, epilogue
writes).  If a compiler merges two of those (`ds_read2_b64`), splits one or moves one, N is too lenient and an
MFMA reads a fragment register whose load is still in flight: no error, wrong pixels.  The hardware does not
interlock, the compiler does not see into the asm; this tool does the count again on the machine code.

It extracts the gfx950 code objects from a built library / object (the `.hip_fatbin` section's offload bundles),
disassembles them (llvm-objdump) and walks every kernel in address order with the queue of outstanding
LGKM-counter operations:
  * `ds_*` push an entry (with the destination registers of the reading forms); scalar memory reads
    (`s_load_*`, `s_buffer_load_*`, `s_memtime`, ...) and `flat_*` push an entry that may return OUT of order;
  * `s_waitcnt ... lgkmcnt(N)` pops the oldest entries until N are left -- unless an out-of-order entry is
    outstanding, in which case only N = 0 proves anything;
  * an instruction that uses a register an outstanding read has yet to deliver is a VIOLATION.
The walk is linear: it follows fall-through, not branches.  That is exact inside the unrolled K loops (straight-
line code, where every counted wait lives) and blind to state arriving over a back edge -- whose targets (layer
loops, tile loops) the kernels enter behind an `lgkmcnt(0)`.  It proves the absence of the failure above in the
code it walks; it is not a model of the whole program.

usage: tools/lds_wait_check.py [library.so | object.o] [kernel-name filter]
"""

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "joshupscale_amd", "lib", "libJoshUpscale.so")


def code_objects(path: str, arch: str = "gfx950"):
    """The device code objects (ELF images) of every offload bundle in `path`'s .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        blob = open(fat, "rb").read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if arch in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(image: bytes) -> str:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name],
                              capture_output=True, text=True, check=True).stdout


_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
_DS_WITH_DST = re.compile(r"^ds_(read|load|bpermute|permute|swizzle|consume|append|ordered_count|.*_rtn)")
_OUT_OF_ORDER = re.compile(r"^(s_load_|s_buffer_load_|s_scratch_load_|s_memtime|s_memrealtime|s_atc_probe|s_dcache_|flat_)")


def regs(text: str):
    found = set()
    for m in _REG.finditer(text):
        if m.group(1):
            found.add((m.group(1), int(m.group(2))))
        else:
            found.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return found


def check_kernel(lines):
    """`lines`: (address text, mnemonic, operand text) of one kernel in address order -> (violations, stats)."""
    queue = []          # outstanding LGKM operations, oldest first: (in_order, destination registers, text, serial number)
    violations = []
    stats = {"ds": 0, "counted_waits": 0, "max_outstanding": 0, "out_of_order_under_counted_wait": 0}
    # The two sides of an if / else on EXEC run on disjoint lanes, and registers are per lane: a read issued on the
    # `then` side is in flight in the then-lanes only, so the else side may overwrite or use "its" register (hipcc
    # does, when it parks an address in an AGPR and reads it back into a fragment register that the else side's own
    # read refills anyway).  The wave's LGKM counter still counts every operation: the queue is unchanged, only the
    # hazard test skips the then-side's entries between `s_andn2_saveexec_b64 D, X` (else: X = the mask an `s_xor_b64 X`
    # computed in front of the then side) and the join `s_or_b64 exec, exec, D`.
    serial = 0
    xor_marks = {}      # SGPR pair text -> serial number at the s_xor_b64 that wrote it
    other_lanes = {}    # join register text -> (first, last) serial numbers issued on the other side
    for addr, mn, ops in lines:
        if mn == "s_xor_b64":
            xor_marks[ops.split(",")[0].strip()] = serial
        elif mn in ("s_andn2_saveexec_b64", "s_or_saveexec_b64"):
            parts = [o.strip() for o in ops.split(",")]
            if len(parts) == 2 and parts[1] in xor_marks:
                other_lanes[parts[0]] = (xor_marks.pop(parts[1]), serial)
        elif mn == "s_or_b64":
            parts = [o.strip() for o in ops.split(",")]
            if len(parts) == 3 and parts[0] == "exec" and parts[1] == "exec":
                other_lanes.pop(parts[2], None)
        if mn == "s_waitcnt":
            m = _LGKM.search(ops)
            if m is None and ops.strip().isdigit():  # (raw immediate: lgkmcnt is bits 11:8)
                n = (int(ops) >> 8) & 0xf
            elif m is None:
                continue
            else:
                n = int(m.group(1))
            if n > 0:
                stats["counted_waits"] += 1
            if n == 0:
                queue.clear()
            elif all(e[0] for e in queue):
                del queue[:max(0, len(queue) - n)]
            elif len(queue) > n:
                stats["out_of_order_under_counted_wait"] += 1   # proves nothing: entries stay
            continue
        if mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # what follows is reached by jumps only, with a state this walk does not know: start afresh (the
            # compiler's own loops, whose loads it waits for itself, are where this happens)
            queue.clear()
            other_lanes.clear()
            continue
        operands = [o.strip() for o in ops.split(",")] if ops else []
        is_ds = mn.startswith("ds_")
        is_ooo = bool(_OUT_OF_ORDER.match(mn))
        has_dst = is_ds and bool(_DS_WITH_DST.match(mn))
        used = regs(", ".join(operands[1:] if has_dst else operands)) if (is_ds or is_ooo) else regs(ops)
        def mine(e):   # (not issued on the other side of an if / else this instruction is inside)
            return not any(lo <= e[3] < hi for lo, hi in other_lanes.values())
        pending = set().union(*(e[1] for e in queue if mine(e))) if queue else set()
        hit = used & pending
        if hit:
            who = [e[2] for e in queue if e[1] & hit and mine(e)]
            violations.append((addr, f"{mn} {ops}", sorted(hit)[:4], who[:2], len(queue)))
        if is_ds:
            stats["ds"] += 1
            queue.append((True, regs(operands[0]) if has_dst and operands else set(), f"{addr} {mn} {ops}", serial))
            serial += 1
        elif is_ooo:
            queue.append((False, set(), f"{addr} {mn} {ops}", serial))
            serial += 1
        stats["max_outstanding"] = max(stats["max_outstanding"], len(queue))
    return violations, stats


def kernels(asm: str):
    """objdump text -> {symbol: [(address, mnemonic, operands)]}"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        addr = line.split("//")[1].strip().split(":")[0] if "//" in line else "?"
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append((addr, parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def check_library(path: str = DEFAULT_LIB, name_filter: str = ""):
    """-> {kernel symbol: (violations, stats)} over every gfx950 code object in `path`."""
    report = {}
    for image in code_objects(path):
        for name, lines in kernels(disassemble(image)).items():
            if name_filter and name_filter not in name:
                continue
            if any(mn.startswith("s_endpgm") for _, mn, _ in lines):   # (kernels and device functions, not data)
                report[name] = check_kernel(lines)
    return report


def main() -> int:
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    report = check_library(path, flt)
    bad = 0
    for name, (violations, stats) in sorted(report.items()):
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"ju::\(anonymous namespace\)::", "", short)[:90]
        print(f"{short:90s} ds {stats['ds']:6d}  counted waits {stats['counted_waits']:5d}  "
              f"max outstanding {stats['max_outstanding']:3d}  violations {len(violations)}")
        for v in violations[:5]:
            print("   ", v)
        bad += len(violations)
    print(f"{len(report)} kernels, {bad} violation(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
