"""Deterministic synthetic read sets for the abyss-bloom-dbg hot path.

Follows the recipe fixed in SURVEY.md section 8d (nothing is bundled with the
reference: README.md:247-250 fetches its test data with wget):

* genome: i.i.d. uniform ACGT (seed 42) with ~1 % of its length in exact
  repeats (500 bp - 5 kbp, 2 - 10 copies) and heterozygous SNPs about every
  1 kbp on a second haplotype;
* pairs: fragment length U[350, 450], both mates ``read_len`` bp, mate 2 is the
  reverse complement of the fragment end, i.i.d. substitution errors, all
  qualities ``I`` (seed 7);
* coverage ``C`` => pairs = G * C / (2 * read_len).

Bases are carried as uint8 codes 0..3 = A, C, G, T (the order of
BASE_CHARS, BloomDBG/RollingBloomDBG.h:26).
"""
from __future__ import annotations

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(length: int, seed: int = 42, repeat_frac: float = 0.01,
                snp_every: int = 1000):
    """Return (hap1, hap2) uint8 code arrays of ``length`` bases."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    # planted exact repeats
    budget = int(length * repeat_frac)
    while budget > 0 and length > 12000:
        rl = int(rng.integers(500, 5001))
        copies = int(rng.integers(2, 11))
        src = int(rng.integers(0, length - rl))
        unit = g[src:src + rl].copy()
        for _ in range(copies - 1):
            dst = int(rng.integers(0, length - rl))
            g[dst:dst + rl] = unit
        budget -= rl * copies
    h2 = g.copy()
    if snp_every > 0:
        nsnp = length // snp_every
        # (without replacement needs a permutation of the whole genome: beyond 2^28 bases the positions
        # are drawn independently instead -- a few coincide, which merely drops those SNPs)
        pos = rng.choice(length, size=nsnp, replace=False) if length <= (1 << 28) else rng.integers(0, length, size=nsnp)
        h2[pos] = (h2[pos] + rng.integers(1, 4, size=nsnp, dtype=np.uint8)) & 3
    return g, h2


def sample_pairs(hap1: np.ndarray, hap2: np.ndarray, n_pairs: int,
                 read_len: int = 150, err: float = 0.005, seed: int = 7,
                 frag_lo: int = 350, frag_hi: int = 450):
    """Return (mate1, mate2) uint8 code matrices [n_pairs, read_len]."""
    rng = np.random.default_rng(seed)
    G = hap1.shape[0]
    frag = rng.integers(frag_lo, frag_hi + 1, size=n_pairs)
    start = rng.integers(0, G - frag_hi, size=n_pairs)
    hap = rng.integers(0, 2, size=n_pairs).astype(bool)
    strand = rng.integers(0, 2, size=n_pairs).astype(bool)
    ar = np.arange(read_len)
    idx1 = start[:, None] + ar[None, :]
    idx2 = (start + frag - 1)[:, None] - ar[None, :]
    m1 = np.where(hap[:, None], hap2[idx1], hap1[idx1])
    m2 = 3 - np.where(hap[:, None], hap2[idx2], hap1[idx2])  # RC of fragment end
    # fragments come from either strand: swap mates' roles
    a = np.where(strand[:, None], m2, m1).astype(np.uint8)
    b = np.where(strand[:, None], m1, m2).astype(np.uint8)
    for m in (a, b):
        e = rng.random(m.shape) < err
        ne = int(e.sum())
        m[e] = (m[e] + rng.integers(1, 4, size=ne, dtype=np.uint8)) & 3
    return a, b


def codes_to_ascii(codes: np.ndarray) -> np.ndarray:
    return BASES[codes]


def write_fastq(path: str, codes: np.ndarray, prefix: str, mate: int) -> None:
    """Write a [n, L] code matrix as 4-line FASTQ with qualities 'I'."""
    n, L = codes.shape
    seqs = codes_to_ascii(codes)
    qual = b"I" * L
    with open(path, "wb") as f:
        chunk = []
        for i in range(n):
            chunk.append(b"@%s%d/%d\n" % (prefix.encode(), i, mate))
            chunk.append(seqs[i].tobytes())
            chunk.append(b"\n+\n")
            chunk.append(qual)
            chunk.append(b"\n")
            if len(chunk) >= 50000:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))


def make_read_set(genome_len: int, coverage: float = 50.0, read_len: int = 150,
                  err: float = 0.005, genome_seed: int = 42, read_seed: int = 7):
    """Genome + pairs at the requested coverage. Returns (mate1, mate2)."""
    h1, h2 = make_genome(genome_len, seed=genome_seed)
    n_pairs = int(genome_len * coverage / (2 * read_len))
    return sample_pairs(h1, h2, n_pairs, read_len=read_len, err=err, seed=read_seed)


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--genome", type=int, default=200000)
    ap.add_argument("--cov", type=float, default=40.0)
    ap.add_argument("--len", type=int, default=150, dest="read_len")
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--gseed", type=int, default=42)
    ap.add_argument("--rseed", type=int, default=7)
    ap.add_argument("--prefix", default="r")
    ap.add_argument("out1")
    ap.add_argument("out2")
    a = ap.parse_args(argv)
    m1, m2 = make_read_set(a.genome, a.cov, a.read_len, a.err, a.gseed, a.rseed)
    write_fastq(a.out1, m1, a.prefix, 1)
    write_fastq(a.out2, m2, a.prefix, 2)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
