#!/usr/bin/env python3
"""Regenerates tests/golden/rresolver/*: the rule of bin/abyss-pe:581-585 run with the UNMODIFIED reference sources --
oracle/_ref/abyss-bloom-dbg -> oracle/_ref/AdjList --dot -> oracle/_ref/abyss-rresolver-short (RResolver/*.cpp compiled
against oracle/shim/btllib/, the restatement of the btllib subset it uses: parity with a real btllib build is unpinned,
see that directory) -- at -j1 on seeded read sets of genomes with planted SHORT exact repeats (a little longer than k, so
that they come out of the unitig stage as repeat unitigs a read spans: what RResolver exists to resolve).

Per case: the reads (npz), the unitigs and their graph (the stage's inputs), and everything the stage writes: the resolved
contigs, the resolved graph and the histograms of -h.  Run in the build container (needs /root/reference):
    python tests/golden/make_rresolver.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from abyss_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "rresolver")

# name, genome length, k, read length, coverage, error rate, repeats (count, shortest, longest, most copies), -b, extra options
CASES = [
    ("rr_k64", 80000, 64, 150, 40.0, 0.003, (40, 66, 116, 4), "16M", []),
    ("rr_k32", 60000, 32, 100, 40.0, 0.003, (40, 34, 70, 3), "8M", []),
    ("rr_k48_t3_x6", 60000, 48, 125, 50.0, 0.005, (30, 50, 100, 4), "8M", ["-t3", "-x6", "-m12", "-M30"]),
]


def genome(G, reps, seed):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=G, dtype=np.uint8)
    n, lo, hi, copies = reps
    for _ in range(n):
        rl = int(rng.integers(lo, hi))
        unit = g[(s := int(rng.integers(0, G - rl))):s + rl].copy()
        for _ in range(int(rng.integers(2, copies + 1)) - 1):
            d = int(rng.integers(0, G - rl))
            g[d:d + rl] = unit
    return g


def run_case(name, G, k, L, cov, err, reps, b, extra, td):
    g = genome(G, reps, 11 + k)
    m1, m2 = synth.sample_pairs(g, g, int(G * cov / (2 * L)), read_len=L, err=err, seed=5 + k,
                                frag_lo=max(2 * L + 10, 350) - 100, frag_hi=max(2 * L + 10, 350))
    seqs = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    with open(os.path.join(td, "reads.fa"), "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">r%d\n%s\n" % (i, s))
    fa = subprocess.run([os.path.join(REF, "abyss-bloom-dbg"), "-k%d" % k, "-b" + b, "-j1", "reads.fa"], cwd=td,
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    open(os.path.join(td, name + "-1.fa"), "wb").write(fa)
    dot = subprocess.run([os.path.join(REF, "AdjList"), "-k%d" % k, "-m%d" % (50 if k > 50 else 0), "--dot", name + "-1.fa"],
                         cwd=td, stdout=subprocess.PIPE, check=True).stdout
    open(os.path.join(td, name + "-1.dot"), "wb").write(dot)
    # (the command of bin/abyss-pe:583-585, -j1 for a deterministic order of the Bloom filter's inserts' side effects: none --
    # the filter is a set -- and of the path tests' histograms)
    cmd = [os.path.join(REF, "abyss-rresolver-short"), "-b" + b, "-f0.8", "-j1", "-k%d" % k] + extra + [
        "-h", name + "-1-rr", "--dot", "-c", name + "-1-rr.fa", "-g", name + "-1-rr.dot", name + "-1.fa", name + "-1.dot", "reads.fa"]
    subprocess.run(cmd, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    np.savez_compressed(os.path.join(OUT, name + ".reads.npz"), buf=np.frombuffer(b"".join(seqs), dtype=np.uint8),
                        off=np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64))
    files = sorted(f for f in os.listdir(td) if f.startswith(name + "-1"))
    for f in files:
        open(os.path.join(OUT, f), "wb").write(open(os.path.join(td, f), "rb").read())
    nin, nout = fa.count(b">"), open(os.path.join(td, name + "-1-rr.fa"), "rb").read().count(b">")
    print(name, "unitigs", nin, "-> contigs", nout, "files", len(files))
    return {"k": k, "bloom": b, "extra": extra, "read_len": L, "reads": len(seqs), "unitigs": nin, "contigs": nout,
            "files": files, "options": " ".join(os.path.basename(c) for c in cmd[1:])}


def main():
    if not os.path.exists(os.path.join(REF, "abyss-rresolver-short")):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for name, G, k, L, cov, err, reps, b, extra in CASES:
        with tempfile.TemporaryDirectory() as td:
            index[name] = run_case(name, G, k, L, cov, err, reps, b, extra, td)
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
