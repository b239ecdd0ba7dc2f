"""Shared by the abyss-rresolver-short tests: the golden cases of tests/golden/rresolver (made by tests/golden/make_rresolver.py
from the unmodified reference) and how to run one of the binaries on them."""
import hashlib
import json
import os
import subprocess

import numpy as np

from util import GOLDEN

RRG = os.path.join(GOLDEN, "rresolver")
INDEX = json.load(open(os.path.join(RRG, "index.json")))
CASES = sorted(k for k in INDEX if not k.startswith("_"))
VARIANTS = INDEX["_variants"]


def digest(data: bytes) -> str:
    """sha256 without the SAM header's @PG line (it names the binary)."""
    return hashlib.sha256(b"".join(l for l in data.splitlines(True) if not l.startswith(b"@PG"))).hexdigest()


def write_inputs(td, name):
    """The unitigs, their graph (dot and adj) and the reads of a case under `td`; returns the read file names."""
    info = INDEX[name]
    for fn in (name + "-1.fa", name + "-1.dot", name + "-1.adj"):
        with open(os.path.join(td, fn), "wb") as f:
            f.write(open(os.path.join(RRG, fn), "rb").read())
    d = np.load(os.path.join(RRG, name + ".reads.npz"))
    buf, off = d["buf"].tobytes(), d["off"]
    seqs = [buf[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    if info.get("read_files") != "fq+fa":
        with open(os.path.join(td, "reads.fa"), "wb") as f:
            f.write(b"".join(b">r%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
        return ["reads.fa"]
    h = len(seqs) // 2
    with open(os.path.join(td, "reads_1.fq"), "wb") as f:
        f.write(b"".join(b"@r%d 1:N:0:x\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs[:h])))
    with open(os.path.join(td, "reads_2.fa"), "wb") as f:
        f.write(b"".join(b">r%d\n%s\n%s\n" % (h + i, s[:60], s[60:]) for i, s in enumerate(seqs[h:])))
    return ["reads_1.fq", "reads_2.fa"]


def run_case(exe, td, name, threads=1, env=None):
    """The rule of bin/abyss-pe:581-585 on a golden case; returns {file: bytes} of what it wrote."""
    info = INDEX[name]
    reads = write_inputs(td, name)
    outs = [f for f in info["files"] if "-1-rr" in f]
    for f in outs:
        if os.path.exists(os.path.join(td, f)):
            os.remove(os.path.join(td, f))
    cmd = [exe, "-b" + info["bloom"], "-f0.8", "-j%d" % threads, "-k%d" % info["k"]] + info["extra"] + [
        "-h", name + "-1-rr", "--dot", "-c", name + "-1-rr.fa", "-g", name + "-1-rr.dot", name + "-1.fa", name + "-1.dot"] + reads
    r = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return {f: open(os.path.join(td, f), "rb").read() for f in outs if os.path.exists(os.path.join(td, f))}


def golden_outputs(name):
    return {f: open(os.path.join(RRG, f), "rb").read() for f in INDEX[name]["files"] if "-1-rr" in f}


def run_variant(exe, td, v, threads=1, env=None):
    """One of the option variants; returns {file: sha256}."""
    info = INDEX[v["base"]]
    reads = write_inputs(td, v["base"])
    base = v["base"]
    cmd = [exe, "-b" + info["bloom"], "-f0.8", "-j%d" % threads, "-k%d" % info["k"]] + info["extra"] + v["extra"] + [
        "-h", "o", "--" + v["format"], "-c", "o.fa", "-g", "o.g", "-S", "o.S", "-U", "o.U", base + "-1.fa", base + "-1." + v["graph_in"]] + reads
    r = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return {f: digest(open(os.path.join(td, f), "rb").read()) for f in sorted(os.listdir(td)) if f.startswith("o")}


def variant_id(v):
    return "%s%s-%s-in_%s" % (v["base"], "".join(v["extra"]), v["format"], v["graph_in"])
