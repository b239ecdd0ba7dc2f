#!/usr/bin/env python3
"""Golden fixtures for the graph shapes the reference's own ExtendPath tests are about (Unittest/Graph/ExtendPathTest.cpp:
cycles, cyclesAndBranches, longestBranch, withTrimming, bidirectional) -- which reads of a random linear genome never make:
circular replicons (a unitig that closes on itself: ER_CYCLE, preprocessCircularContig, bloom-dbg.h:648-702), tandem repeats
with units shorter and longer than k (cycles with a way in and a way out), inverted repeats and hairpins (a k-mer next to
its own reverse complement), homopolymers and dinucleotide runs (a vertex that is its own neighbour).

Same layout as make_golden.py: the UNMODIFIED reference (oracle/_ref/abyss-bloom-dbg, `make -C oracle ref`) at -j1 on seeded
reads; per case the reads, the unitig FASTA, the --read-log, the -T trace (without its `length` column) and the counting
filter's statistics.  Only runs where /root/reference exists; the fixtures are committed."""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "abyss-bloom-dbg")
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def rnd(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))


def rc(s):
    return s.translate(COMP)[::-1]


def replicons(kind, rng):
    """[(sequence, circular)]"""
    if kind == "plasmids":
        # circles shorter than a read, about as long as one, and long; one of them twice the same unit (a cycle inside a cycle)
        unit = rnd(rng, 170)
        return [(rnd(rng, 90), True), (rnd(rng, 260), True), (rnd(rng, 2100), True), (rnd(rng, 7000), True),
                (unit + unit, True), (rnd(rng, 3000), False)]
    if kind == "tandem":
        g = [rnd(rng, 1500)]
        for unit, copies in ((7, 30), (20, 10), (33, 8), (50, 6), (120, 4), (400, 3)):
            g += [rnd(rng, unit) * copies, rnd(rng, 1200)]
        return [(b"".join(g), False)]
    if kind == "inverted":
        a, b, c = rnd(rng, 600), rnd(rng, 45), rnd(rng, 300)
        g = [rnd(rng, 1500), a, rnd(rng, 900), rc(a), rnd(rng, 1200),          # an inverted repeat 900 bp apart
             b, rc(b), rnd(rng, 1000),                                           # a perfect hairpin (palindrome of 90 bp)
             c, rnd(rng, 10), rc(c), rnd(rng, 1500),                             # a hairpin with a 10-base loop
             b"ACGT" * 20, rnd(rng, 800), b"GAATTC" * 12, rnd(rng, 1500)]        # runs of short palindromes
        return [(b"".join(g), False)]
    if kind == "lowcomplex":
        g = [rnd(rng, 1200), b"A" * 80, rnd(rng, 700), b"AC" * 50, rnd(rng, 700), b"T" * 45 + b"G" + b"T" * 45, rnd(rng, 900),
             b"AAG" * 30, rnd(rng, 600), b"C" * 30, rnd(rng, 1200)]
        return [(b"".join(g), False)]
    if kind == "satellite":
        # a 10 kbp homopolymer and a 10 kbp dinucleotide satellite at 500-fold coverage beside an ordinary 8 kbp sequence at 40-fold
        # (third field: a coverage multiplier): their k-mers recur tens of thousands of times in a batch of PASS 1 -- more pairs
        # on one counter than a tile's bin holds -- and everything about them saturates (counters at 255)
        return [(rnd(rng, 8000), False), (rnd(rng, 400) + b"A" * 10000 + rnd(rng, 400), False, 12.5), (rnd(rng, 400) + b"AC" * 5000 + rnd(rng, 400), False, 12.5)]
    if kind == "mixed":
        # (the k-mer size at its limits: MAX_KMER = 192 wants long reads; a small k makes a crowded graph)
        return [(rnd(rng, 9000), False), (rnd(rng, 1400), True), (rnd(rng, 60) * 12, False)]
    raise ValueError(kind)


def sample(reps, rng, cov, L, err):
    reads = []
    for rep in reps:
        seq, circ = rep[0], rep[1]
        n = max(4, int(len(seq) * cov * (rep[2] if len(rep) > 2 else 1.0) / L))
        if circ:
            ext = seq * (L // len(seq) + 2)
            starts = rng.integers(0, len(seq), size=n)
        else:
            ext = seq
            starts = rng.integers(0, max(1, len(seq) - L + 1), size=n)
        for s in starts:
            r = bytearray(ext[int(s):int(s) + L])
            for p in np.nonzero(rng.random(len(r)) < err)[0]:
                r[p] = b"ACGT"[(b"ACGT".index(r[p]) + int(rng.integers(1, 4))) & 3]
            r = bytes(r)
            reads.append(rc(r) if rng.random() < 0.5 else r)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


# name, kind, coverage, read length, error rate, reference options
CASES = [
    ("s_plasmids_k32", "plasmids", 40.0, 120, 0.004, ["-k32", "-b2M"]),
    ("s_tandem_k32", "tandem", 40.0, 120, 0.004, ["-k32", "-b2M"]),
    ("s_tandem_k64_t20", "tandem", 40.0, 150, 0.004, ["-k64", "-b2M", "-t20"]),
    ("s_inverted_k40", "inverted", 40.0, 120, 0.004, ["-k40", "-b2M"]),
    ("s_lowcomplex_k25", "lowcomplex", 40.0, 100, 0.004, ["-k25", "-b2M"]),
    ("s_plasmids_k48_K16", "plasmids", 40.0, 120, 0.004, ["-k48", "-K16", "-b2M"]),
    ("s_mixed_k192", "mixed", 60.0, 250, 0.003, ["-k192", "-b2M"]),
    ("s_mixed_k12", "mixed", 30.0, 100, 0.004, ["-k12", "-b2M"]),
    # the number of hash functions away from the usual four: one (nothing left for the second probe stage), six (beyond the
    # four-at-a-time paths of PASS 1), twelve (beyond the eight a cooperative probe round holds)
    ("s_mixed_k32_H1", "mixed", 30.0, 120, 0.004, ["-k32", "-b2M", "-H1"]),
    ("s_mixed_k40_H6", "mixed", 30.0, 120, 0.004, ["-k40", "-b2M", "-H6"]),
    ("s_mixed_k32_H12_kc3", "mixed", 40.0, 120, 0.004, ["-k32", "-b3M", "-H12", "--kc=3"]),
    ("s_satellite_k40", "satellite", 40.0, 120, 0.004, ["-k40", "-b4M"]),
]


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    for ci, (name, kind, cov, L, err, opts) in enumerate(CASES):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        rng = np.random.default_rng(1000 + ci)
        seqs = sample(replicons(kind, rng), rng, cov, L, err)
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "reads.fa"), "wb") as f:
                for i, s in enumerate(seqs):
                    f.write(b">r%d\n%s\n" % (i, s))
            r = subprocess.run([REF] + opts + ["-j1", "-v", "--read-log=rl.tsv", "-T", "tr.tsv", "reads.fa"],
                               cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            errtxt = r.stderr.decode()
            stats = {"counters": int(re.search(r"#counters\s+= (\d+)", errtxt).group(1)),
                     "filtered_popcount": int(re.search(r"popcount\s+= (\d+)", errtxt).group(1)), "options": opts, "kind": kind}
            m = re.search(r"Processed (\d+) reads, solid reads: (\d+) .*visited reads: (\d+)", errtxt.splitlines()[-3])
            stats.update(reads=int(m.group(1)), solid_reads=int(m.group(2)), visited_reads=int(m.group(3)))
            open(os.path.join(HERE, name + ".fa"), "wb").write(r.stdout)
            open(os.path.join(HERE, name + ".readlog.tsv"), "wb").write(open(os.path.join(td, "rl.tsv"), "rb").read())
            rows = [ln.split("\t") for ln in open(os.path.join(td, "tr.tsv")).read().splitlines()]
            with open(os.path.join(HERE, name + ".trace.tsv"), "w") as f:
                for row in rows:
                    f.write("\t".join(row[:1] + row[2:]) + "\n")
            stats["unitigs"] = r.stdout.count(b">")
            json.dump(stats, open(os.path.join(HERE, name + ".json"), "w"), indent=1, sort_keys=True)
        np.savez_compressed(os.path.join(HERE, name + ".reads.npz"), buf=np.frombuffer(b"".join(seqs), dtype=np.uint8),
                            off=np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64))
        print(name, stats)


if __name__ == "__main__":
    main()
