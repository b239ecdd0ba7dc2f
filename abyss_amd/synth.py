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


# ---------------------------------------------------------------------------------------------
# Counter-based twin of sample_pairs: every random draw is a pure function of (seed, pair, draw)
# or (seed, read, base) through splitmix64, so the SAME read set can be produced on the host
# (numpy, to write the FASTQ files the reference binary reads) and on the GPU (torch, where
# bench.py times it already resident in HBM), bit for bit, without carrying 3 GB of files to the
# GPU box.  tests/test_synth.py checks the two against each other.
_SM_GAMMA = 0x9E3779B97F4A7C15
_SM_M1 = 0xBF58476D1CE4E5B9
_SM_M2 = 0x94D049BB133111EB
_MASK64 = (1 << 64) - 1


def _splitmix_np(x):
    """splitmix64 finaliser of a uint64 array (wraps modulo 2^64)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(_SM_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_SM_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_SM_M2)
        return z ^ (z >> np.uint64(31))


def _i64(c: int) -> int:
    """A 64-bit constant as the signed value torch.int64 holds for the same bits."""
    return c - (1 << 64) if c >= (1 << 63) else c


def _splitmix_torch(x):
    """The same on a torch.int64 tensor: int64 arithmetic wraps like uint64, shifts are made logical."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    z = x + _i64(_SM_GAMMA)
    z = (z ^ lsr(z, 30)) * _i64(_SM_M1)
    z = (z ^ lsr(z, 27)) * _i64(_SM_M2)
    return z ^ lsr(z, 31)


def pair_draws(n_pairs: int, genome_len: int, seed: int, first: int = 0, frag_lo: int = 350, frag_hi: int = 450, xp="numpy", device=None):
    """(frag, start, hap, strand) of pairs [first, first + n_pairs): 53 high bits of one splitmix64 word each."""
    if xp == "numpy":
        i = np.arange(first, first + n_pairs, dtype=np.uint64)
        with np.errstate(over="ignore"):
            base = np.uint64((seed * 0xD1342543DE82EF95) & _MASK64) + i * np.uint64(4)
            u = [_splitmix_np(base + np.uint64(j)) >> np.uint64(11) for j in range(3)]
        frag = (u[0] % np.uint64(frag_hi - frag_lo + 1)).astype(np.int64) + frag_lo
        start = (u[1] % np.uint64(genome_len - frag_hi)).astype(np.int64)
        return frag, start, (u[2] & np.uint64(1)).astype(bool), ((u[2] >> np.uint64(1)) & np.uint64(1)).astype(bool)
    import torch
    i = torch.arange(first, first + n_pairs, dtype=torch.int64, device=device)
    base = _i64((seed * 0xD1342543DE82EF95) & _MASK64) + i * 4
    u = [(_splitmix_torch(base + j) >> 11) & ((1 << 53) - 1) for j in range(3)]
    frag = u[0] % (frag_hi - frag_lo + 1) + frag_lo
    start = u[1] % (genome_len - frag_hi)
    return frag, start, (u[2] & 1).bool(), ((u[2] >> 1) & 1).bool()


def error_draws(first_read: int, n_reads: int, read_len: int, err: float, seed: int, xp="numpy", device=None):
    """Per base of reads [first_read, first_read + n_reads): (substitute?, offset 1..3).  A base is
    substituted when the low 24 bits of its word fall below err * 2^24."""
    thr = int(round(err * (1 << 24)))
    if xp == "numpy":
        r = np.arange(first_read, first_read + n_reads, dtype=np.uint64)[:, None]
        c = np.arange(read_len, dtype=np.uint64)[None, :]
        with np.errstate(over="ignore"):
            w = _splitmix_np(np.uint64((seed * 0xA24BAED4963EE407 + 0x5851F42D4C957F2D) & _MASK64) + r * np.uint64(read_len) + c)
        return (w & np.uint64(0xFFFFFF)) < np.uint64(thr), ((w >> np.uint64(24)) % np.uint64(3)).astype(np.uint8) + 1
    import torch
    r = torch.arange(first_read, first_read + n_reads, dtype=torch.int64, device=device)[:, None]
    c = torch.arange(read_len, dtype=torch.int64, device=device)[None, :]
    w = _splitmix_torch(_i64((seed * 0xA24BAED4963EE407 + 0x5851F42D4C957F2D) & _MASK64) + r * read_len + c)
    return (w & 0xFFFFFF) < thr, (((w >> 24) & ((1 << 40) - 1)) % 3).to(torch.uint8) + 1


def sample_pairs_cb(hap1: np.ndarray, hap2: np.ndarray, n_pairs: int, read_len: int = 150, err: float = 0.005,
                    seed: int = 7, first: int = 0, total_pairs: int = None):
    """sample_pairs with counter-based draws (numpy).  Reads are numbered as in the FASTQ files the
    reference is given -- all mate-1 reads, then all mate-2 reads (BloomIO.h:102-115) -- so read
    `total_pairs + i` is the mate of read `i`; `first` / `total_pairs` let a caller make a slice."""
    total_pairs = n_pairs if total_pairs is None else total_pairs
    G = hap1.shape[0]
    frag, start, hap, strand = pair_draws(n_pairs, G, seed, first)
    ar = np.arange(read_len)
    idx1 = start[:, None] + ar[None, :]
    idx2 = (start + frag - 1)[:, None] - ar[None, :]
    m1 = np.where(hap[:, None], hap2[idx1], hap1[idx1])
    m2 = 3 - np.where(hap[:, None], hap2[idx2], hap1[idx2])
    a = np.where(strand[:, None], m2, m1).astype(np.uint8)
    b = np.where(strand[:, None], m1, m2).astype(np.uint8)
    for m, r0 in ((a, first), (b, total_pairs + first)):
        e, off = error_draws(r0, n_pairs, read_len, err, seed)
        m[e] = (m[e] + off[e]) & 3
    return a, b


def make_read_set_cb(genome_len: int, coverage: float = 50.0, read_len: int = 150, err: float = 0.005,
                     genome_seed: int = 42, read_seed: int = 7):
    """make_read_set with counter-based pair sampling: the read set bench.py generates on the GPU."""
    h1, h2 = make_genome(genome_len, seed=genome_seed)
    n_pairs = int(genome_len * coverage / (2 * read_len))
    return sample_pairs_cb(h1, h2, n_pairs, read_len=read_len, err=err, seed=read_seed)


def packed_reads_torch(hap1: np.ndarray, hap2: np.ndarray, n_pairs: int, read_len: int, err: float, seed: int, device,
                       first: int = 0, total_pairs: int = None, chunk: int = 1 << 19):
    """The reads of sample_pairs_cb generated on `device` in the packed layout of include/abyss_amd.h:
    2 bits per base, 16 bases per uint32 word, each read on a word boundary; all mate-1 reads of the
    slice, then all its mate-2 reads.  Returns (words, woff, lens) tensors."""
    import torch
    total_pairs = n_pairs if total_pairs is None else total_pairs
    G = hap1.shape[0]
    g1 = torch.from_numpy(hap1).to(device)
    g2 = torch.from_numpy(hap2).to(device)
    wpr = (read_len + 15) // 16
    words = torch.empty((2 * n_pairs, wpr), dtype=torch.int32, device=device)
    shifts = 2 * torch.arange(16, device=device, dtype=torch.int64)
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    for a0 in range(0, n_pairs, chunk):
        m = min(n_pairs, a0 + chunk) - a0
        frag, start, hap, strand = pair_draws(m, G, seed, first + a0, xp="torch", device=device)
        hap, strand = hap[:, None], strand[:, None]
        idx1 = start[:, None] + ar[None, :]
        idx2 = (start + frag - 1)[:, None] - ar[None, :]
        m1 = torch.where(hap, g2[idx1], g1[idx1])
        m2 = 3 - torch.where(hap, g2[idx2], g1[idx2])
        for dst, r, r0 in ((a0, torch.where(strand, m2, m1), first + a0), (n_pairs + a0, torch.where(strand, m1, m2), total_pairs + first + a0)):
            e, off = error_draws(r0, m, read_len, err, seed, xp="torch", device=device)
            r = torch.where(e, (r + off) & 3, r)
            r64 = torch.nn.functional.pad(r.to(torch.int64), (0, wpr * 16 - read_len)).view(m, wpr, 16)
            words[dst:dst + m] = (r64 << shifts).sum(dim=2).to(torch.int32)  # wraps modulo 2^32: same bits as uint32
    n = 2 * n_pairs
    woff = torch.arange(n + 1, device=device, dtype=torch.int64) * wpr
    lens = torch.full((n,), read_len, dtype=torch.int32, device=device)
    return words.view(-1), woff, lens


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
    ap.add_argument("--cb", action="store_true", help="counter-based pair sampling (make_read_set_cb): the read set of bench.py")
    ap.add_argument("out1")
    ap.add_argument("out2")
    a = ap.parse_args(argv)
    m1, m2 = (make_read_set_cb if a.cb else make_read_set)(a.genome, a.cov, a.read_len, a.err, a.gseed, a.rseed)
    write_fastq(a.out1, m1, a.prefix, 1)
    write_fastq(a.out2, m2, a.prefix, 2)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
