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
