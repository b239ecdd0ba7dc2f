#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Runs oracle/_ref/abyss-bloom-dbg (built by `make -C oracle ref` from the sources under
/root/reference; see oracle/Makefile) at -j1 on seeded synthetic read sets and stores,
per case: the reads (2-bit codes, npz), the unitig FASTA, the --read-log, the -T trace
(without the `length` column, which the reference leaves uninitialised for redundant
contigs) and the counting-filter statistics it prints with -v.  Also stores ntHash
streams and a raw counter array produced by the reference's own nthash.hpp /
CountingBloomFilter.hpp through oracle/_ref/tier1.

Only runs where /root/reference exists (this container); the fixtures are committed so
that the GPU box, which has no reference sources, can check against them.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abyss_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(ROOT, "oracle", "_ref", "abyss-bloom-dbg")
TIER1 = os.path.join(ROOT, "oracle", "_ref", "tier1")

# name, genome length, coverage, read length, reference options, decorate (Ns / lower case / short reads)
CASES = [
    ("k32", 20000, 30.0, 150, ["-k32", "-b4M"], False),
    ("k64", 20000, 30.0, 150, ["-k64", "-b4M"], False),
    ("k25_h3_kc3_t40", 20000, 30.0, 150, ["-k25", "-b3M", "-H3", "--kc=3", "-t40"], False),
    ("k40_mixed", 30000, 30.0, 100, ["-k40", "-b6M"], True),
    ("k96", 30000, 40.0, 150, ["-k96", "-b6M"], False),
    ("k48_K16", 20000, 30.0, 150, ["-k48", "-K16", "-b4M"], False),
    ("k50_qr11", 20000, 30.0, 150, ["-k50", "--qr-seed=11", "-b4M"], False),
]


def decorate(asc, rng):
    """Ns, lower-case stretches and a few reads shorter than k, as FastaReader would pass them on."""
    seqs = [bytes(r) for r in asc]
    n = len(seqs)
    for i in rng.choice(n, size=n // 50, replace=False):
        s = bytearray(seqs[i]); s[int(rng.integers(0, len(s)))] = ord("N"); seqs[i] = bytes(s)
    for i in rng.choice(n, size=n // 100, replace=False):
        seqs[i] = seqs[i][: int(rng.integers(5, 39))]
    return seqs


def write_fasta(path, seqs, prefix):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">%s%d\n%s\n" % (prefix.encode(), i, s))


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    for name, G, cov, L, opts, deco in CASES:
        m1, m2 = synth.make_read_set(G, cov, read_len=L)
        codes = np.concatenate([m1, m2])
        asc = synth.codes_to_ascii(codes)
        seqs = [bytes(r) for r in asc]
        if deco:
            seqs = decorate(asc, np.random.default_rng(1))
        with tempfile.TemporaryDirectory() as td:
            write_fasta(os.path.join(td, "reads.fa"), seqs, "r")
            r = subprocess.run([REF] + opts + ["-j1", "-v", "--read-log=rl.tsv", "-T", "tr.tsv", "reads.fa"],
                               cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            err = r.stderr.decode()
            stats = {
                "counters": int(re.search(r"#counters\s+= (\d+)", err).group(1)),
                "filtered_popcount": int(re.search(r"popcount\s+= (\d+)", err).group(1)),
                "options": opts,
            }
            m = re.search(r"Processed (\d+) reads, solid reads: (\d+) .*visited reads: (\d+)", err.splitlines()[-3])
            stats.update(reads=int(m.group(1)), solid_reads=int(m.group(2)), visited_reads=int(m.group(3)))
            open(os.path.join(HERE, name + ".fa"), "wb").write(r.stdout)
            open(os.path.join(HERE, name + ".readlog.tsv"), "wb").write(open(os.path.join(td, "rl.tsv"), "rb").read())
            rows = [ln.split("\t") for ln in open(os.path.join(td, "tr.tsv")).read().splitlines()]
            with open(os.path.join(HERE, name + ".trace.tsv"), "w") as f:
                for row in rows:
                    f.write("\t".join(row[:1] + row[2:]) + "\n")
            json.dump(stats, open(os.path.join(HERE, name + ".json"), "w"), indent=1, sort_keys=True)
        # reads as stored sequences (lengths vary when decorated)
        np.savez_compressed(os.path.join(HERE, name + ".reads.npz"),
                            buf=np.frombuffer(b"".join(seqs), dtype=np.uint8),
                            off=np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64))
        print(name, stats)
    # ntHash streams and a raw counter array from the reference's own headers
    rng = np.random.default_rng(3)
    seqs = ["ACGTACACTGGACTGAGTCT"] + ["".join(rng.choice(list("ACGT"), size=int(n))) for n in (40, 77, 130, 200)]
    seqs.append("ACGTTGCATGNNACGTAGCTAGCTAGGATCGATTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTACGNACGATCGATCGATCGACTAGCT")
    vectors = []
    for k in (5, 20, 32, 33, 64, 65, 96):
        for s in seqs:
            if len(s) < k:
                continue
            out = subprocess.run([TIER1, "hash", str(k), "4", s], stdout=subprocess.PIPE, check=True).stdout.decode()
            rows = [[int(x) for x in ln.split()] for ln in out.splitlines()]
            vectors.append({"k": k, "seq": s, "pos": [r[0] for r in rows], "hashes": [[str(v) for v in r[1:]] for r in rows]})
    json.dump(vectors, open(os.path.join(HERE, "nthash_vectors.json"), "w"))
    lines = "\n".join("".join(rng.choice(list("ACGT"), size=120)) for _ in range(300))
    lines = lines + "\n" + lines[: 120 * 40]  # repeats so that counters exceed 1
    m, k, H = 8192, 24, 4
    r = subprocess.run([TIER1, "counters", str(m), str(k), str(H), "2"], input=lines.encode(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, check=True)
    np.savez_compressed(os.path.join(HERE, "tier1_counters.npz"), counters=np.frombuffer(r.stdout, dtype=np.uint8),
                        lines=np.frombuffer(lines.encode(), dtype=np.uint8), m=m, k=k, H=H)
    print("tier1:", r.stderr.decode().strip())
    # Bloom file formats + cascading filter (abyss-bloom build -t counting / -t rolling-hash -l 2)
    m1, m2 = synth.make_read_set(20000, 30.0)
    reads = "\n".join(bytes(r).decode() for r in synth.codes_to_ascii(np.concatenate([m1, m2])))
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([TIER1, "counters", "262144", "32", "3", "0", os.path.join(td, "c.bloom")], input=reads.encode(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        open(os.path.join(HERE, "file_counting_k32_h3.bloom"), "wb").write(open(os.path.join(td, "c.bloom"), "rb").read())
        r = subprocess.run([TIER1, "cascade", str(1 << 20), "32", "2", "2", os.path.join(td, "b.bloom")], input=reads.encode(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        open(os.path.join(HERE, "file_cascade_k32_h2_l2.bloom"), "wb").write(open(os.path.join(td, "b.bloom"), "rb").read())
        r = subprocess.run([TIER1, "cascade", str(1 << 16), "24", "3", "3", os.path.join(td, "b3.bloom")], input=reads.encode(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        open(os.path.join(HERE, "file_cascade_k24_h3_l3.bloom"), "wb").write(open(os.path.join(td, "b3.bloom"), "rb").read())
    print("bloom files written")


if __name__ == "__main__":
    main()
