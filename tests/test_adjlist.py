"""AdjList (the stage after the unitigs) on the CPU: the oracle restatement
against the unmodified reference, and the product's host side (adjlist_core.h) over the product's
join logic run serially by tests/hostcheck, against both.  The GPU twin is test_gpu_adjlist.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import adjlist_oracle as ao
from abyss_amd import api, build
from util import GOLDEN

ADJ_GOLDEN = os.path.join(GOLDEN, "adjlist")
INDEX = json.load(open(os.path.join(ADJ_GOLDEN, "index.json")))
FORMATS = ["adj", "dot", "gfa1", "gfa2", "asqg", "sam"]


def strip_pg(data: bytes) -> bytes:
    return b"".join(l for l in data.splitlines(True) if not l.startswith(b"@PG"))


def golden_outputs(name):
    return {fmt: open(os.path.join(ADJ_GOLDEN, "%s.%s" % (name, fmt)), "rb").read()
            for fmt in FORMATS if os.path.exists(os.path.join(ADJ_GOLDEN, "%s.%s" % (name, fmt)))}


def synthetic_contigs(seed, k, n=400, genome=30000, short_every=7):
    """A contig set with every kind of end the join has to tell apart: consecutive pieces of a
    random genome sharing exactly k-1 bases (or fewer: the suffix-array overlaps), in random
    orientation; repeated pieces (several vertices with one end); homopolymer and palindromic
    ends; lower case and ambiguity codes inside; a contig of the minimum length k."""
    rng = np.random.default_rng(seed)
    g = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), genome))
    recs, at, i = [], 0, 0
    while len(recs) < n and at + 4 * k < genome:
        ln = int(rng.integers(k, 3 * k))
        piece = g[at:at + ln]
        if rng.random() < 0.5:
            piece = ao.revcomp(piece)
        if rng.random() < 0.2 and ln > 2 * k + 4:
            mid = ln // 2
            piece = piece[:mid] + bytes([b"acgtNRYKMSWBDHV"[int(rng.integers(0, 15))]]) + piece[mid + 1:]
        recs.append(("%d" % i, "%d %d" % (ln, int(rng.integers(0, 5000))), piece))
        i += 1
        back = k - 1 if (i % short_every) else int(rng.integers(1, k - 1))
        at += ln - back
        if rng.random() < 0.05:
            recs.append(("%d" % i, "%d 7" % ln, recs[-1][2]))  # the same contig again
            i += 1
    pal = (b"ACGT" * k)[:k - 1]
    recs.append(("polyA", "%d 3" % (2 * k), b"A" * (2 * k)))
    recs.append(("polyT", "", b"T" * (k + 3)))
    recs.append(("pal", "x y", pal + b"GATTACA" + ao.revcomp(pal)))
    recs.append(("mink", "%d 1" % k, g[5:5 + k]))
    return recs


def write_fasta(path, recs, width=0):
    with open(path, "wb") as f:
        for rid, comment, seq in recs:
            f.write(b">" + rid.encode() + ((b" " + comment.encode()) if comment else b"") + b"\n")
            if width:
                for a in range(0, len(seq), width):
                    f.write(seq[a:a + width] + b"\n")
            else:
                f.write(seq + b"\n")


def run_bin(binary, k, m, fmt, extra, fasta):
    r = subprocess.run([binary, "-k%d" % k, "-m%d" % m, "--" + fmt] + list(extra) + [fasta], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    return strip_pg(r.stdout)


@pytest.fixture(scope="module")
def adjlist_check():
    build.build_hostcheck()
    return build.ADJLIST_CHECK


@pytest.mark.parametrize("name", sorted(INDEX))
def test_oracle_restatement_matches_the_reference_outputs(name):
    c = INDEX[name]
    recs = ao.read_fasta(os.path.join(GOLDEN, c["fasta"]))
    contigs, out = ao.build(recs, c["k"], c["m"], "--SS" in c["extra"])
    gold = golden_outputs(name)
    assert ao.format_adj(contigs, out) == gold["adj"]
    assert ao.format_dot(contigs, out) == gold["dot"]


@pytest.mark.parametrize("name", sorted(INDEX))
def test_host_side_over_the_serial_join_writes_the_reference_outputs(name, adjlist_check):
    c = INDEX[name]
    gold = golden_outputs(name)
    assert set(gold) >= {"adj", "dot"}
    for fmt, want in gold.items():
        got = run_bin(adjlist_check, c["k"], c["m"], fmt, c["extra"], os.path.join(GOLDEN, c["fasta"]))
        assert got == want, (name, fmt)


@pytest.mark.parametrize("k,m,ss,seed", [(21, 0, False, 1), (33, 10, False, 2), (33, 10, True, 3), (64, 50, False, 4),
                                         (65, 20, True, 5), (97, 50, False, 6), (130, 60, False, 7), (250, 200, False, 8)])
def test_synthetic_contig_sets(k, m, ss, seed, adjlist_check, tmp_path):
    recs = synthetic_contigs(seed, k)
    fa = str(tmp_path / "contigs.fa")
    write_fasta(fa, recs, width=60 if seed % 2 else 0)
    contigs, out = ao.build(recs, k, m, ss)
    extra = ["--SS"] if ss else []
    assert sum(len(x) for x in out) > len(recs) // 2
    if m and m < k - 1:
        assert any(d != -(k - 1) for x in out for _, d in x)  # (the set does hold shorter overlaps)
    got_adj = run_bin(adjlist_check, k, m, "adj", extra, fa)
    assert got_adj == ao.format_adj(contigs, out)
    assert run_bin(adjlist_check, k, m, "dot", extra, fa) == ao.format_dot(contigs, out)
    if os.path.exists(ao.REF_ADJLIST) and k - 1 <= 128:  # (the reference build's MAX_KMER)
        for fmt in FORMATS:
            assert run_bin(adjlist_check, k, m, fmt, extra, fa) == run_bin(ao.REF_ADJLIST, k, m, fmt, extra, fa), fmt


def hc_join(overlap, head, tail, ss):
    l = C.CDLL(build.build_hostcheck())
    l.hc_overlap_join.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    n = head.shape[0]
    off = np.zeros(2 * n + 1, dtype=np.uint64)
    ne = C.c_uint64()
    l.hc_overlap_join(overlap, n, head.ctypes.data, tail.ctypes.data, int(ss), off.ctypes.data, None, C.byref(ne))
    tgt = np.zeros(max(1, ne.value), dtype=np.uint32)
    l.hc_overlap_join(overlap, n, head.ctypes.data, tail.ctypes.data, int(ss), off.ctypes.data, tgt.ctypes.data, C.byref(ne))
    return off, tgt[:ne.value]


def csr_of(out):
    off = np.zeros(len(out) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in out])
    return off, np.array([v for x in out for v, _ in x], dtype=np.uint32)


@pytest.mark.parametrize("overlap,ss", [(1, False), (2, True), (3, False), (31, False), (32, True), (33, False), (64, False),
                                        (65, True), (200, False), (256, False)])
def test_join_logic_on_keys_that_collide(overlap, ss):
    # ends drawn from a handful of strings: long adjacency lists, every pair of senses, equal keys
    rng = np.random.default_rng(overlap)
    pool = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), overlap)) for _ in range(6)]
    pool += [ao.revcomp(pool[0]), b"A" * overlap, b"T" * overlap, (b"AT" * overlap)[:overlap]]
    recs = []
    for i in range(300):
        h, t = pool[int(rng.integers(len(pool)))], pool[int(rng.integers(len(pool)))]
        recs.append(("%d" % i, "", h + b"C" + t))
    contigs = ao.Contigs(recs, overlap + 1)
    want_off, want_tgt = csr_of(ao.overlap_edges(contigs, ss))
    head, tail = api.pack_ends([r[2] for r in recs], overlap)
    off, tgt = hc_join(overlap, head, tail, ss)
    assert np.array_equal(off, want_off) and np.array_equal(tgt, want_tgt)
    assert int(off[-1]) > 3000


def test_join_of_nothing_and_of_one():
    off, tgt = hc_join(5, np.zeros((0, 1), np.uint64), np.zeros((0, 1), np.uint64), False)
    assert off.tolist() == [0] and len(tgt) == 0
    head, tail = api.pack_ends([b"ACGTTTTTTACGT"], 4)  # ACGT is its own reverse complement
    off, tgt = hc_join(4, head, tail, False)
    contigs = ao.Contigs([("0", "", b"ACGTTTTTTACGT")], 5)
    want_off, want_tgt = csr_of(ao.overlap_edges(contigs))
    assert off.tolist() == want_off.tolist() and tgt.tolist() == want_tgt.tolist() and len(tgt) == 4


def test_host_side_errors_like_the_reference(adjlist_check, tmp_path):
    fa = str(tmp_path / "n.fa")
    write_fasta(fa, [("0", "", b"ACGTACGTNA")])
    r = subprocess.run([adjlist_check, "-k4", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"unexpected character: 'N'" in r.stderr  # Common/Sequence.cpp:101-104
    write_fasta(fa, [("0", "", b"ACGT")])
    r = subprocess.run([adjlist_check, "-k6", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"longer than k-1" in r.stderr  # AdjList.cpp:209 (an assertion there)
    r = subprocess.run([adjlist_check, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"missing -k,--kmer option" in r.stderr  # AdjList.cpp:367-370
    r = subprocess.run([adjlist_check, "-k5x", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"invalid option: `-k5x'" in r.stderr  # AdjList.cpp:360-364


def test_verbose_statistics_match_the_reference(adjlist_check):
    if not os.path.exists(ao.REF_ADJLIST):
        pytest.skip("oracle/_ref/AdjList is not built here")
    for fa, k, m in (("k32.fa", 32, 0), ("k96.fa", 96, 20)):
        outs = []
        for b in (ao.REF_ADJLIST, adjlist_check):
            r = subprocess.run([b, "-v", "-k%d" % k, "-m%d" % m, os.path.join(GOLDEN, fa)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0
            outs.append(r.stderr)
        assert outs[0] == outs[1]


def test_reads_stdin_and_several_files(adjlist_check, tmp_path):
    recs = synthetic_contigs(11, 25, n=60)
    a, b = str(tmp_path / "a.fa"), str(tmp_path / "b.fa")
    write_fasta(a, recs[:30])
    write_fasta(b, recs[30:])
    contigs, out = ao.build(recs, 25, 0)
    r = subprocess.run([adjlist_check, "-k25", "-m0", a, b], stdout=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == ao.format_adj(contigs, out)
    r = subprocess.run([adjlist_check, "-k25", "-m0"], input=open(a, "rb").read() + open(b, "rb").read(), stdout=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == ao.format_adj(contigs, out)


def test_random_small_contig_sets_against_the_reference_binary(adjlist_check, tmp_path):
    """Ends drawn from tiny pools over small alphabets (long adjacency lists, palindromes, contigs shorter than
    2(k-1) whose ends overlap), ambiguity codes inside, odd comments, every format, --SS, many k and m."""
    if not os.path.exists(ao.REF_ADJLIST):
        pytest.skip("oracle/_ref/AdjList is not built here")
    import random
    rnd = random.Random(5)
    fa = str(tmp_path / "fuzz.fa")
    for case in range(80):
        k = rnd.choice([4, 5, 8, 16, 21, 31, 32, 33, 34, 48, 64, 65, 66, 96, 97, 127, 128, 129])
        alpha = rnd.choice([b"ACGT", b"AC", b"A", b"ACGT" * 3 + b"acgt"])
        pool = [bytes(rnd.choice(alpha) for _ in range(k - 1)) for _ in range(rnd.randint(1, 6))]
        recs = []
        for i in range(rnd.randint(1, 40)):
            h, t = rnd.choice(pool), rnd.choice(pool)
            if rnd.random() < 0.3:
                t = ao.revcomp(h.upper())
            if rnd.random() < 0.2:
                s = bytes(rnd.choice(alpha) for _ in range(rnd.randint(k, max(k, 2 * k - 2))))
            else:
                s = h + bytes(rnd.choice(b"ACGTNRYKMSWBDHVacgtn") for _ in range(rnd.randint(0, 5))) + t
            s += b"A" * max(0, k - len(s))
            recs.append(("c%d" % i, rnd.choice(["", "%d %d" % (len(s), rnd.randint(0, 999)), "x", "12", "5 9 extra"]), s))
        # (-m1 is left out: the reference's chop() asserts on its last, one-base query whenever a contig is blunt)
        m = rnd.choice([0, 2, max(2, k // 2), max(2, k - 2), k - 1, 50, 1000])
        extra = ["--SS"] if rnd.random() < 0.3 else []
        write_fasta(fa, recs, width=rnd.choice([0, 0, 7, 60]))
        for fmt in FORMATS:
            assert run_bin(adjlist_check, k, m, fmt, extra, fa) == run_bin(ao.REF_ADJLIST, k, m, fmt, extra, fa), (case, k, m, extra, fmt)


def test_contigs_parsed_block_parallel_change_nothing(adjlist_check, tmp_path, monkeypatch):
    """read_fasta_blocks (a plain FASTA file of a megabyte or more is cut at record starts and parsed by several threads), forced
    on small files: the goldens, wrapped and unwrapped synthetic sets with ambiguity codes -- and the reader's complaints, which
    must name the same line."""
    outs = {}
    for force in ("-1", "1"):  # (-1: never)
        monkeypatch.setenv("ABG_FASTA_BLOCKS_MIN", force)
        for name in sorted(INDEX):
            c = INDEX[name]
            for fmt in ("adj", "dot"):
                outs[(force, name, fmt)] = run_bin(adjlist_check, c["k"], c["m"], fmt, c["extra"], os.path.join(GOLDEN, c["fasta"]))
        for seed, width in ((21, 0), (22, 60), (23, 7)):
            recs = synthetic_contigs(seed, 33, n=900)
            recs = [(i, c, s.replace(b"AC", b"MY", 1) if n % 5 == 0 else s) for n, (i, c, s) in enumerate(recs)]
            fa = str(tmp_path / ("c%d.fa" % seed))
            write_fasta(fa, recs, width=width)
            outs[(force, seed, "adj")] = run_bin(adjlist_check, 33, 10, "adj", [], fa)
        # an empty record in the middle: FastaReader's message with the line number
        recs = synthetic_contigs(31, 25, n=300)
        fa = str(tmp_path / "broken.fa")
        write_fasta(fa, recs[:150] + [("x", "", b"")] + recs[150:])
        r = subprocess.run([adjlist_check, "-k25", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0 and b"is empty" in r.stderr
        outs[(force, "broken", "stderr")] = r.stderr
    for (force, a, b), v in outs.items():
        if force == "1":
            assert v == outs[("-1", a, b)], (a, b)
