"""Build helpers: compile the HIP library (product) and the checkers (test infrastructure).

The product is one shared library, ``abyss_amd/lib/libabyss_amd.so`` (plus the host binaries over its C ABI in ``abyss_amd/bin``), built in-tree with
``hipcc --offload-arch=gfx950`` from ``abyss_amd/csrc/abg_kernels.hip``; hipcc
cross-compiles without a GPU.  The checkers (``oracle/liboracle.so``, ``oracle/abg_oracle``,
``oracle/_ref/*`` when /root/reference is present, ``tests/hostcheck/libhostcheck.so``) are
only ever loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "abyss_amd", "csrc")
LIB = os.path.join(ROOT, "abyss_amd", "lib", "libabyss_amd.so")
BIN_DIR = os.path.join(ROOT, "abyss_amd", "bin")
ORACLE_DIR = os.path.join(ROOT, "oracle")
HOSTCHECK = os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so")
REFERENCE = "/root/reference"


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: %s" % " ".join(cmd))
    return r.stdout


def hipcc() -> str:
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found")


def csrc_files():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "abyss_amd.h")]


OBJ_DIR = os.path.join(ROOT, "abyss_amd", "lib", "obj")
# translation units of the library and what each is rebuilt for (the big one takes minutes; the others seconds)
UNITS = {
    "abg_kernels": ["abg_kernels.hip", "abg_core.h", "abg_engine.h", "abg_walk.h", "abg_host.h", "abg_overlap.h"],
    "abg_rr": ["abg_rr.hip", "abg_rr.h", "abg_core.h"],
}


def build_lib(force: bool = False) -> str:
    """libabyss_amd.so: the gfx950 kernels + C ABI, one object per translation unit."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    header = os.path.join(ROOT, "include", "abyss_amd.h")
    objs, procs = [], []
    for unit, deps in UNITS.items():
        obj = os.path.join(OBJ_DIR, unit + ".o")
        objs.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, d) for d in deps] + [header]):
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", obj, os.path.join(CSRC, unit + ".hip")]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("build step failed: %s" % " ".join(cmd))
    if force or procs or _newer(LIB, objs):
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_cli(force: bool = False) -> str:
    """abyss_amd/bin/abyss-bloom-dbg: the drop-in host binary (C++ over the C ABI)."""
    src = os.path.join(CSRC, "host", "bloom_dbg_main.cc")
    out = os.path.join(BIN_DIR, "abyss-bloom-dbg")
    if not os.path.exists(src):
        return ""
    deps = [src] + [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    if force or _newer(out, deps + [LIB]):
        os.makedirs(BIN_DIR, exist_ok=True)
        build_lib()
        _run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-o", out, src,
              "-L" + os.path.dirname(LIB), "-labyss_amd", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread"])
        _run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-o",
              os.path.join(BIN_DIR, "abyss-bloom"), os.path.join(CSRC, "host", "bloom_main.cc"),
              "-L" + os.path.dirname(LIB), "-labyss_amd", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread"])
        _run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-o",
              os.path.join(BIN_DIR, "AdjList"), os.path.join(CSRC, "host", "adjlist_main.cc"),
              "-L" + os.path.dirname(LIB), "-labyss_amd", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread"])
        _run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-o",
              os.path.join(BIN_DIR, "abyss-rresolver-short"), os.path.join(CSRC, "host", "rresolver_main.cc"),
              "-L" + os.path.dirname(LIB), "-labyss_amd", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread"])
    return out


def build_oracle(force: bool = False) -> None:
    """C restatement (always) and the unmodified reference (when its sources are present)."""
    targets = ["oracle"]
    if os.path.isdir(REFERENCE):
        targets.append("ref")
    if force:
        _run(["make", "-C", ORACLE_DIR, "clean"])
    _run(["make", "-C", ORACLE_DIR, "-j8"] + targets)


READER_CHECK = os.path.join(ROOT, "tests", "hostcheck", "reader_check")
ADJLIST_CHECK = os.path.join(ROOT, "tests", "hostcheck", "adjlist_check")
RRESOLVER_CHECK = os.path.join(ROOT, "tests", "hostcheck", "rresolver_check")


def build_hostcheck(force: bool = False) -> str:
    src = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cc")
    if force or _newer(HOSTCHECK, [src] + csrc_files()):
        _run(["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", HOSTCHECK, src])
    rsrc = os.path.join(ROOT, "tests", "hostcheck", "reader_check.cc")
    if force or _newer(READER_CHECK, [rsrc, os.path.join(CSRC, "host", "fasta_reader.h")]):
        _run(["g++", "-std=c++17", "-O2", "-o", READER_CHECK, rsrc])
    asrc = os.path.join(ROOT, "tests", "hostcheck", "adjlist_check.cc")
    if force or _newer(ADJLIST_CHECK, [asrc, HOSTCHECK, os.path.join(CSRC, "host", "adjlist_core.h"),
                                       os.path.join(CSRC, "host", "fasta_reader.h")]):
        _run(["g++", "-std=c++17", "-O2", "-o", ADJLIST_CHECK, asrc, "-L" + os.path.dirname(HOSTCHECK), "-lhostcheck",
              "-Wl,-rpath,$ORIGIN"])
    hsrc = [os.path.join(CSRC, "host", f) for f in ("rresolver_core.h", "graph_writers.h", "fasta_reader.h", "si_bytes.h")]
    csrc = os.path.join(ROOT, "tests", "hostcheck", "rresolver_check.cc")
    if force or _newer(RRESOLVER_CHECK, [csrc, os.path.join(CSRC, "abg_rr.h"), os.path.join(CSRC, "abg_core.h")] + hsrc):
        _run(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-o", RRESOLVER_CHECK, csrc, "-lpthread"])
    return HOSTCHECK


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_cli(force)
    build_oracle(force)
    build_hostcheck(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", LIB)
