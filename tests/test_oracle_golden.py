"""The oracle (oracle/abg_oracle.c) against the reference's own vectors and outputs.

Golden fixtures come from the unmodified reference (tests/golden/make_golden.py); the only
known-answer vector in the reference tree is vendor/nthash/unittest/UnitTests.cpp:39-53.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api
from util import GOLDEN, GoldenCase, mask_of

ALL_CASES = ["k32", "k64", "k25_h3_kc3_t40", "k40_mixed", "k96", "k48_K16", "k50_qr11"]


def test_nthash_known_answer():
    # vendor/nthash/unittest/UnitTests.cpp:39-53
    o = ob.Oracle(20, counters=1024, num_hashes=3)
    pos, h = o.hash_seq(b"ACGTACACTGGACTGAGTCT")
    assert list(pos) == [0]
    assert [int(x) for x in h[0]] == [10434435546371013747, 16073887395445158014, 8061578976118370557]


def test_nthash_streams_match_reference_headers():
    vectors = json.load(open(os.path.join(GOLDEN, "nthash_vectors.json")))
    assert len(vectors) > 20
    for v in vectors:
        o = ob.Oracle(v["k"], counters=1024, num_hashes=4)
        pos, h = o.hash_seq(v["seq"].encode())
        assert list(pos) == v["pos"], (v["k"], v["seq"])
        assert [[str(int(x)) for x in row] for row in h] == v["hashes"]


def test_counter_array_matches_reference_filter():
    z = np.load(os.path.join(GOLDEN, "tier1_counters.npz"))
    lines = z["lines"].tobytes().split(b"\n")
    buf, off = api.concat_seqs(lines)
    o = ob.Oracle(int(z["k"]), counters=int(z["m"]), num_hashes=int(z["H"]))
    o.load(buf, off)
    assert np.array_equal(o.counters(), z["counters"])


def test_filter_sizing_matches_reference():
    # bloom-dbg.cc:365-367 as logged by the reference for these budgets
    for name in ("k32", "k25_h3_kc3_t40", "k96"):
        g = GoldenCase(name)
        assert ob.lib().orc_counters_for_budget(g.opts["bloom_bytes"]) == g.meta["counters"]
    assert ob.lib().orc_counters_for_budget(2 << 30) == 1908874368
    assert ob.lib().orc_counters_for_budget(100 << 20) == 93206784


def _mask_of(g):
    buf = C.create_string_buffer(g.opts["k"] + 1)
    if "K" in g.opts:
        ob.lib().orc_seed_kmer_pair(g.opts["k"], g.opts["K"], buf)
        return buf.value
    if "qr" in g.opts:
        ob.lib().orc_seed_qr_pair(g.opts["k"], g.opts["qr"], buf)
        return buf.value
    return None


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_reproduces_reference_run(name):
    g = GoldenCase(name)
    o = ob.Oracle(mask=_mask_of(g), **g.kwargs())
    assert o.size == g.meta["counters"]
    o.load(g.buf, g.off)
    assert o.counting_stats()[1] == g.meta["filtered_popcount"]
    results, contigs = o.assemble(g.buf, g.off)
    assert api.format_fasta(contigs, g.ids) == g.fasta
    assert api.format_read_log(results, g.ids) == g.readlog
    assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace
    c = o.assembly_counters()
    assert c["reads_processed"] == g.meta["reads"]
    assert c["solid_reads"] == g.meta["solid_reads"]
    assert c["visited_reads"] == g.meta["visited_reads"]


def test_spaced_seed_strings():
    # Unittest/BloomDBG/SpacedSeedTest.cpp:16,25
    buf = C.create_string_buffer(64)
    ob.lib().orc_seed_qr(11, buf)
    assert buf.value == b"10100011101"
    ob.lib().orc_seed_qr_pair(33, 11, buf)
    assert buf.value == b"101000111010000000000010111000101"
    ob.lib().orc_seed_kmer_pair(10, 3, buf)
    assert buf.value == b"1110000111"


def test_rolling_hash_identities():
    # Unittest/BloomDBG/RollingHashTest.cpp:34-45,178-199: rolling == reset; reverse complement
    # gives the same canonical hash stream in reverse order
    o = ob.Oracle(13, counters=1024, num_hashes=2)
    s = b"GCAATGTTAGCCATTACGGATTGCAAACTGACCGGTTA"
    pos, h = o.hash_seq(s)
    for p in pos:
        _, h1 = o.hash_seq(s[p:p + 13])
        assert np.array_equal(h1[0], h[p])
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    _, hr = o.hash_seq(s.translate(comp)[::-1])
    assert np.array_equal(hr[::-1], h)


def test_iterator_skips_non_acgt():
    # Unittest/BloomDBG/RollingHashIteratorTest.cpp:64-104
    o = ob.Oracle(4, counters=1024, num_hashes=1)
    pos, _ = o.hash_seq(b"ACGTNACGTACGNACG")
    assert list(pos) == [0, 5, 6, 7, 8]
    assert len(o.hash_seq(b"ACG")[0]) == 0
    assert len(o.hash_seq(b"")[0]) == 0
    pos2, h2 = o.hash_seq(b"acgtnacgt")
    assert list(pos2) == [0, 5]


GRAPH_GOLDEN = json.load(open(os.path.join(GOLDEN, "graph_golden.json")))


STRUCTURED = ["s_plasmids_k32", "s_tandem_k32", "s_tandem_k64_t20", "s_inverted_k40", "s_lowcomplex_k25", "s_plasmids_k48_K16", "s_mixed_k192", "s_mixed_k12", "s_mixed_k32_H1", "s_mixed_k40_H6", "s_mixed_k32_H12_kc3", "s_satellite_k40"]


@pytest.mark.parametrize("name", STRUCTURED)
def test_oracle_reproduces_reference_run_on_cycles_repeats_and_hairpins(name):
    """Graph shapes a random linear genome never makes (tests/golden/make_structured.py, from the unmodified reference at -j1):
    circular replicons, tandem repeats with units shorter and longer than k, inverted repeats and hairpins, homopolymer and
    dinucleotide runs -- what Unittest/Graph/ExtendPathTest.cpp's cycles / cyclesAndBranches / longestBranch / withTrimming
    cases are about, as k-mer graphs."""
    test_oracle_reproduces_reference_run(name)


@pytest.mark.parametrize("name", ["k32", "k64", "k40_mixed", "k48_K16", "k25_h3_kc3_t40"])
def test_oracle_graphviz_dump_matches_reference(name):
    """-g (outputGraph, bloom-dbg.h:1171-1242): the oracle's GraphViz text against the SHA-256 / size /
    visitor counters of the file the unmodified reference wrote for the same reads
    (tests/golden/make_graph_golden.py)."""
    import hashlib
    g = GoldenCase(name)
    kw = g.kwargs()
    m = mask_of(g)
    o = ob.Oracle(kw["k"], counters=g.meta["counters"], num_hashes=kw["num_hashes"], min_cov=kw["min_cov"],
                  mask=m.encode() if m else None)
    o.load(g.buf, g.off)
    text, nodes, edges = o.output_graph(g.buf, g.off)
    ref = GRAPH_GOLDEN[name]
    assert (len(text), nodes, edges) == (ref["bytes"], ref["nodes"], ref["edges"])
    assert hashlib.sha256(text).hexdigest() == ref["sha256"]


def test_oracle_graphviz_dump_small_case_in_full():
    import gzip
    g = GoldenCase("k32")
    kw = g.kwargs()
    o = ob.Oracle(kw["k"], counters=g.meta["counters"])
    buf, off = api.concat_seqs(g.reads[:300])
    o.load(buf, off)
    text, nodes, edges = o.output_graph(buf, off)
    ref = gzip.open(os.path.join(GOLDEN, "k32_first300.graph.dot.gz"), "rb").read()
    assert text == ref
    assert (nodes, edges) == (GRAPH_GOLDEN["k32_first300"]["nodes"], GRAPH_GOLDEN["k32_first300"]["edges"])
    assert text.startswith(b"digraph g {\n\t") and text.endswith(b";\n}\n")


# ---- the reference's own unit tests for this path, restated against the oracle
TOY = [b"CGACT", b"TGACT", b"GACTC", b"ACTCT", b"ACTCG"]  # Unittest/BloomDBG/RollingBloomDBGTest.cpp:31-57
BASE_BIT = {"A": 1, "C": 2, "G": 4, "T": 8}


def _toy_graph(mask=None, num_hashes=2):
    o = ob.Oracle(5, counters=100000, num_hashes=num_hashes, min_cov=1, mask=mask)
    buf, off = api.concat_seqs(TOY)
    o.load(buf, off)
    return o


def test_reference_unit_test_toy_graph_neighbours():
    """RollingBloomDBGTest out_edge_iterator / adjacency_iterator / in_edges / pathTraversal
    (Unittest/BloomDBG/RollingBloomDBGTest.cpp:61-191): GACTC has successors ACTCT, ACTCG and
    predecessors CGACT, TGACT; CGACT -> GACTC and GACTC -> ACTCG/ACTCT are the only edges on the way."""
    o = _toy_graph()
    assert o.out_mask(b"GACTC") == BASE_BIT["T"] | BASE_BIT["G"]
    assert o.in_mask(b"GACTC") == BASE_BIT["C"] | BASE_BIT["T"]
    assert o.out_mask(b"CGACT") == BASE_BIT["C"]          # out_degree(CGACT) == 1: -> GACTC
    assert o.out_mask(b"ACTCG") == 0 and o.out_mask(b"ACTCT") == 0
    assert o.in_mask(b"CGACT") == 0 and o.in_mask(b"TGACT") == 0


def test_reference_unit_test_toy_graph_under_a_spaced_seed():
    """RollingBloomDBGSpacedSeedTest (:231-340): seed 11011, one hash; GACTC equals its own reverse
    complement under the mask, which must not add edges."""
    o = _toy_graph(mask=b"11011", num_hashes=1)
    assert o.out_mask(b"GACTC") == BASE_BIT["T"] | BASE_BIT["G"]
    assert o.in_mask(b"GACTC") == BASE_BIT["C"] | BASE_BIT["T"]


def test_reference_unit_test_counting_filter_threshold():
    """CountingBloomFilter base (Unittest/BloomDBG/CountingBloomFilterTest.cpp:9-47): 1000 counters, one
    hash, threshold 2 -- contains() and filtered_popcount() after each insert."""
    k, a, b, c, d, e = 16, b"AGATGTGCTGCCGCCT", b"TGGACAGCGTTACCTC", b"TAATAACAGTCCCTAT", b"GATCGTGGCGGGCGAT", b"T" * 16
    o = ob.Oracle(k, counters=1000, num_hashes=1, min_cov=2)
    assert o.size == 1000

    def insert(s):
        buf, off = api.concat_seqs([s])
        o.load(buf, off)

    def contains(s):
        return bool(o.min_count(o.hash_seq(s)[1])[0] >= 2)

    insert(a)
    assert o.counting_stats()[1] == 0 and not contains(e)
    insert(a)
    assert o.counting_stats()[1] == 1 and contains(a)
    insert(b)
    assert o.counting_stats()[1] == 1 and not contains(b)
    insert(c)
    assert o.counting_stats()[1] == 1 and not contains(c)
    insert(b)
    assert o.counting_stats()[1] == 2 and contains(b) and not contains(d)


def test_reference_unit_test_path_to_seq_under_a_spaced_seed():
    """BloomDBG pathToSeq (Unittest/BloomDBG/BloomDBGTest.cpp:21-38): ACGTAC as a path of two 5-mers under
    the seed 10001 comes back as ACNNAC.  Reached through the assembly: a filter holding both k-mers twice
    makes the read solid, and the contig of that read is the path's sequence."""
    seq = b"ACGTAC"
    o = ob.Oracle(5, counters=100000, num_hashes=2, min_cov=1, trim=0, mask=b"10001")
    # flanks so that the read has no blunt end (5 solid predecessors / successors either side)
    long = b"GGCATTCAGCA" + seq + b"GTCTTGACCAT"
    buf, off = api.concat_seqs([long])
    o.load(buf, off)
    res, contigs = o.assemble(buf, off)
    assert len(contigs) >= 1
    text = b"".join(c.seq for c in contigs)
    assert b"N" in text  # columns no '1' of the seed covers come back as N (bloom-dbg.h:130-158)
    k = 5
    for c in contigs:  # every column covered by the first or last position of some path k-mer is a base
        assert all(ch in b"ACGTN" for ch in c.seq)
        assert c.seq[0] in b"ACGT" and c.seq[-1] in b"ACGT"
