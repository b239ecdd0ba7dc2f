"""Randomised parameter sweep of the device logic (serial host execution) against the oracle:
odd and even k, saturated filters (heavy false-positive branching), trim 0..2k, 1..9 hash
functions, tiny claim tables and insert batches."""
import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api, synth
from test_hostcheck import HostCheck
from util import contig_tuple


@pytest.mark.parametrize("trial", range(10))
def test_random_configuration(trial):
    rng = np.random.default_rng(1000 + trial)
    G = int(rng.integers(3000, 9000))
    k = int(rng.choice([9, 12, 15, 16, 21, 24, 31, 32, 33, 40, 48, 63, 64, 65, 80]))
    cov = float(rng.choice([15, 25]))
    err = float(rng.choice([0.005, 0.02, 0.05]))
    kc = int(rng.choice([1, 2, 3]))
    trim = int(rng.choice([0, 1, 3, k // 2, k, 2 * k]))
    H = int(rng.choice([1, 2, 4, 5, 9]))
    h1, h2 = synth.make_genome(G, seed=trial, snp_every=int(rng.choice([0, 100, 500])))
    m1, m2 = synth.sample_pairs(h1, h2, int(G * cov / 200), read_len=100, err=err, seed=trial + 100, frag_lo=150, frag_hi=250)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    counters = int(rng.choice([1 << 17, 1 << 19, 99991 * 8]))
    o = ob.Oracle(k, counters=counters, num_hashes=H, min_cov=kc, trim=trim)
    hc = HostCheck(k, counters, H, kc, trim, insert_batch=int(rng.choice([3000, 50000])), claim_log2=int(rng.choice([8, 14])),
                   p2_first=int(rng.choice([16, 256, 100000])))
    o.load(buf, off)
    hc.load(buf, off)
    assert np.array_equal(o.counters(), hc.counters())
    ro, co = o.assemble(buf, off)
    rh, ch = hc.assemble(buf, off)
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert np.array_equal(o.visited(), hc.visited())
    assert o.assembly_counters() == hc.assembly_counters()


def _structured_genome(rng, G):
    """A random sequence with what arc_ends_decided's conditions are about planted in it every few hundred bases: tandem repeats
    with units of 1 .. 8 bases, perfect and near-perfect hairpins (a stem, a short loop, the stem's reverse complement), and
    stretches that are their own reverse complement."""
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    rc = lambda s: bytes(comp[c] for c in reversed(s))
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), size=int(n)).astype(np.uint8))
    out = []
    while sum(map(len, out)) < G:
        out.append(rnd(rng.integers(120, 400)))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            unit = rnd(rng.integers(1, 9))
            out.append((unit * 200)[:int(rng.integers(30, 140))])
        elif kind == 1:
            stem = rnd(rng.integers(15, 80))
            out.append(stem + rnd(rng.integers(0, 6)) + rc(stem))
        elif kind == 2:
            half = rnd(rng.integers(10, 60))
            out.append(half + rc(half))  # (even length, its own reverse complement)
        else:
            stem = bytearray(rnd(rng.integers(20, 70)))
            arm = bytearray(rc(bytes(stem)))
            arm[int(rng.integers(0, len(arm)))] = int(rng.choice(list(b"ACGT")))  # (one mismatch in the stem)
            out.append(bytes(stem) + rnd(rng.integers(0, 4)) + bytes(arm))
    return b"".join(out)[:G]


@pytest.mark.parametrize("trial", range(12))
def test_archive_verdicts_where_the_look_ahead_argument_is_at_its_edge(trial, monkeypatch):
    """Round 6: a read well inside an archived contig gets both of its blunt-end look-aheads answered without a search, where a period
    <= 5 or a reverse-palindromic overlap of the end k-mer with the contig beyond it cannot have blocked the contig's own path
    (arc_ends_decided).  Genomes made of exactly such structure -- short-unit tandem repeats, hairpins, self-complementary stretches --
    at many k, in many small batches so that most reads meet the archive: every read's verdict, every contig and the visited filter
    as the oracle's sequential run leaves them, and as the same engine leaves them with the archive off."""
    rng = np.random.default_rng(7000 + trial)
    G = int(rng.integers(4000, 9000))
    k = int(rng.choice([20, 21, 24, 27, 31, 32, 33, 40, 47, 48]))
    g = _structured_genome(rng, G)
    L = 100
    n = int(G * 30 / L)
    starts = rng.integers(0, G - L, size=n)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for i, s in enumerate(starts):
        r = bytearray(g[s:s + L])
        for j in np.nonzero(rng.random(L) < 0.003)[0]:
            r[j] = int(rng.choice(list(b"ACGT")))
        r = bytes(r)
        reads.append(r if i % 2 == 0 else r.translate(comp)[::-1])
    buf, off = api.concat_seqs(reads)
    counters = 1 << 20
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    decided = {}
    for arch in ("1", "0"):
        monkeypatch.setenv("ABG_CLS_ARCHIVE", arch)
        hc = HostCheck(k, counters, insert_batch=50000, claim_log2=14, p2_first=int(rng.choice([32, 64, 128])))
        hc.load(buf, off)
        rh, ch = hc.assemble(buf, off)
        decided[arch] = hc.stats()["cls_decided_reads"]
        assert np.array_equal(ro, rh), (trial, k, arch, int(np.sum(ro != rh)))
        assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch], (trial, k, arch)
        assert np.array_equal(o.visited(), hc.visited())
    assert decided["0"] == 0 and decided["1"] > 0, decided
