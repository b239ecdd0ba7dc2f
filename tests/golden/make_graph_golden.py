#!/usr/bin/env python3
"""Golden vectors for `-g` (outputGraph, BloomDBG/bloom-dbg.h:1171-1242) from the unmodified
reference binary (oracle/_ref/abyss-bloom-dbg, built by `make -C oracle ref`): for the read sets
of the existing golden cases, the SHA-256, size and node / edge counts of the GraphViz file the
reference writes, plus one small file in full (the first 300 reads of k32).  Run in the build
container (needs /root/reference); the fixtures travel, the reference does not."""
import gzip
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from util import GoldenCase  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "abyss-bloom-dbg")


def run(g, reads, opts):
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "reads.fa"), "wb") as f:
            for i, r in enumerate(reads):
                f.write(b">r%d\n%s\n" % (i, r))
        r = subprocess.run([REF] + opts + ["-j1", "-v", "-g", "g.dot", "reads.fa"], cwd=td, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, check=True)
        dot = open(os.path.join(td, "g.dot"), "rb").read()
        m = re.search(r"processed \d+ reads \(k-mers visited: (\d+), edges visited: (\d+)\)", r.stderr.decode())
        return dot, int(m.group(1)), int(m.group(2))


def main():
    out = {}
    for name in ("k32", "k64", "k40_mixed", "k48_K16", "k25_h3_kc3_t40"):
        g = GoldenCase(name)
        dot, nodes, edges = run(g, g.reads, g.meta["options"])
        out[name] = {"sha256": hashlib.sha256(dot).hexdigest(), "bytes": len(dot), "nodes": nodes, "edges": edges}
        print(name, out[name])
    g = GoldenCase("k32")
    dot, nodes, edges = run(g, g.reads[:300], g.meta["options"])
    out["k32_first300"] = {"sha256": hashlib.sha256(dot).hexdigest(), "bytes": len(dot), "nodes": nodes, "edges": edges}
    with gzip.GzipFile(os.path.join(HERE, "k32_first300.graph.dot.gz"), "wb", mtime=0) as f:
        f.write(dot)
    json.dump(out, open(os.path.join(HERE, "graph_golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
