#!/usr/bin/env python3
"""What the compiler made of the kernels, from the in-tree library alone (no GPU needed): per device function of
libabyss_amd.so's gfx950 code object -- instructions, loads, loads that are waited for on the spot (the next memory
instruction is `s_waitcnt vmcnt(0)`: a round trip that nothing overlaps), moves of spilled scalars (v_readlane /
v_writelane) and scratch accesses.  See notes/isa_lens.md for what to do about them.

usage: python tools/isa_report.py [pattern ...]     (patterns match the mangled names; default: the NW = 2 build)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    td = tempfile.mkdtemp(prefix="isa_")
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()


def main():
    pats = sys.argv[1:] or ["ILi2E"]
    lines = disassemble(os.path.join(ROOT, "abyss_amd", "lib", "libabyss_amd.so"))
    starts = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^[0-9a-f]+ <(.*)>:$", ln)] if m]
    load = re.compile(r"\b(flat_load|global_load|buffer_load)")
    rows = []
    for n, (a, name) in enumerate(starts):
        if not any(p in name for p in pats):
            continue
        body = lines[a + 1:starts[n + 1][0] if n + 1 < len(starts) else len(lines)]
        loads = [j for j, x in enumerate(body) if load.search(x)]
        waited = 0
        for j in loads:
            for x in body[j + 1:j + 6]:
                if "s_waitcnt vmcnt(0)" in x:
                    waited += 1
                    break
                if load.search(x):
                    break
        rows.append((len(body), len(loads), waited, sum("v_readlane" in x or "v_writelane" in x for x in body),
                     sum("scratch_" in x for x in body), name))
    print("%8s %6s %7s %7s %8s  %s" % ("instrs", "loads", "waited", "spills", "scratch", "function"))
    for r in sorted(rows, reverse=True):
        print("%8d %6d %7d %7d %8d  %s" % (r[0], r[1], r[2], r[3], r[4], r[5][:120]))


if __name__ == "__main__":
    main()
