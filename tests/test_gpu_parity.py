"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden
outputs, on a real MI355X.  Bit-exact everywhere: hashes, counters, visited bits, read
results, unitig FASTA (ids, coverage, order) and trace."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api, synth
from util import GOLDEN, GoldenCase, contig_tuple, mask_of

pytestmark = pytest.mark.gpu


def test_hash_stream_matches_reference_vectors():
    vectors = json.load(open(os.path.join(GOLDEN, "nthash_vectors.json")))
    by_k = {}
    for v in vectors:
        by_k.setdefault(v["k"], []).append(v)
    for k, vs in by_k.items():
        g = api.BloomDBG(k, counters=4096)
        for v in vs:
            pos, h = g.hash_seq(v["seq"].encode())
            assert list(pos) == v["pos"]
            assert [[str(int(x)) for x in row] for row in h] == v["hashes"]
        g.close()


def test_counter_array_matches_reference_filter():
    z = np.load(os.path.join(GOLDEN, "tier1_counters.npz"))
    buf, off = api.concat_seqs(z["lines"].tobytes().split(b"\n"))
    g = api.BloomDBG(int(z["k"]), counters=int(z["m"]), num_hashes=int(z["H"]))
    g.load(buf, off)
    assert np.array_equal(g.counters(), z["counters"])


@pytest.mark.parametrize("name", ["k32", "k64", "k25_h3_kc3_t40", "k40_mixed", "k96", "k48_K16", "k50_qr11"])
def test_reproduces_reference_run(name):
    gc = GoldenCase(name)
    g = api.BloomDBG(spaced_seed=mask_of(gc), **gc.kwargs())
    assert g.size == gc.meta["counters"]
    g.load(gc.buf, gc.off)
    assert g.counting_stats()[1] == gc.meta["filtered_popcount"]
    results, contigs = g.assemble(gc.buf, gc.off)
    assert api.format_fasta(contigs, gc.ids) == gc.fasta
    assert api.format_read_log(results, gc.ids) == gc.readlog
    assert api.format_trace(contigs, gc.ids, gc.reads, gc.opts["k"], with_length=False) == gc.trace
    c = g.assembly_counters()
    assert (c["reads_processed"], c["solid_reads"], c["visited_reads"]) == (
        gc.meta["reads"], gc.meta["solid_reads"], gc.meta["visited_reads"])
    # the accelerators ran, whatever the seed and the parity of k: read-guided bulk steps and chains, the memo
    st = g.stats()
    assert st["bulk_steps"] > st["lin_steps"] and st["chain_steps"] > 0 and st["memo_hits"] > 0, (name, st)


@pytest.mark.parametrize("name", ["s_plasmids_k32", "s_tandem_k32", "s_tandem_k64_t20", "s_inverted_k40", "s_lowcomplex_k25", "s_plasmids_k48_K16", "s_mixed_k192", "s_mixed_k12", "s_mixed_k32_H1", "s_mixed_k40_H6", "s_mixed_k32_H12_kc3", "s_satellite_k40"])
def test_reproduces_reference_run_on_cycles_repeats_and_hairpins(name):
    """Graph shapes a random linear genome never makes (tests/golden/make_structured.py, from the unmodified reference at -j1):
    circular replicons, tandem repeats with units shorter and longer than k, inverted repeats and hairpins, homopolymer and
    dinucleotide runs -- what Unittest/Graph/ExtendPathTest.cpp's cycles / cyclesAndBranches / longestBranch / withTrimming
    cases are about, as k-mer graphs."""
    gc = GoldenCase(name)
    g = api.BloomDBG(spaced_seed=mask_of(gc), **gc.kwargs())
    assert g.size == gc.meta["counters"]
    g.load(gc.buf, gc.off)
    assert g.counting_stats()[1] == gc.meta["filtered_popcount"]
    results, contigs = g.assemble(gc.buf, gc.off)
    assert api.format_fasta(contigs, gc.ids) == gc.fasta
    assert api.format_read_log(results, gc.ids) == gc.readlog
    assert api.format_trace(contigs, gc.ids, gc.reads, gc.opts["k"], with_length=False) == gc.trace
    c = g.assembly_counters()
    assert (c["reads_processed"], c["solid_reads"], c["visited_reads"]) == (
        gc.meta["reads"], gc.meta["solid_reads"], gc.meta["visited_reads"])
    g.close()


@pytest.mark.parametrize("k,G,cov", [(21, 40000, 30.0), (64, 200000, 40.0), (33, 60000, 30.0), (97, 50000, 40.0),
                                     (150, 20000, 30.0)])
def test_matches_oracle(k, G, cov):
    m1, m2 = synth.make_read_set(G, cov)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    counters = 1 << 24
    o = ob.Oracle(k, counters=counters)
    g = api.BloomDBG(k, counters=counters)
    o.load(buf, off)
    g.load(buf, off)
    assert np.array_equal(o.counters(), g.counters())
    assert o.counting_stats() == g.counting_stats()
    ro, co = o.assemble(buf, off)
    rg, cg = g.assemble(buf, off)
    assert np.array_equal(ro, rg)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    assert np.array_equal(o.visited(), g.visited())
    assert o.assembly_counters() == g.assembly_counters()


@pytest.mark.parametrize("trial,k", [(0, 32), (1, 41), (2, 64)])
def test_archive_verdicts_on_repeats_and_hairpins_match_oracle(trial, k, monkeypatch):
    """The structure arc_ends_decided's conditions are about (tests/test_stress_logic.py: short-unit tandem repeats, hairpins,
    self-complementary stretches), on the GPU in many small batches: verdicts, contigs and the visited filter are the oracle's."""
    from test_stress_logic import _structured_genome
    rng = np.random.default_rng(9000 + trial)
    G, L = 60000, 120
    g = _structured_genome(rng, G)
    starts = rng.integers(0, G - L, size=int(G * 30 / L))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for i, s in enumerate(starts):
        r = bytearray(g[s:s + L])
        for j in np.nonzero(rng.random(L) < 0.003)[0]:
            r[j] = int(rng.choice(list(b"ACGT")))
        r = bytes(r)
        reads.append(r if i % 2 == 0 else r.translate(comp)[::-1])
    buf, off = api.concat_seqs(reads)
    monkeypatch.setenv("ABG_P2_FIRST_BATCH", "256")
    counters = 1 << 22
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    gg = api.BloomDBG(k, counters=counters)
    gg.load(buf, off)
    rg, cg = gg.assemble(buf, off)
    st = gg.stats()
    assert st["cls_decided_reads"] > 0 and st["walk_rounds"] > 3, st
    assert np.array_equal(ro, rg), int(np.sum(ro != rg))
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    assert np.array_equal(o.visited(), gg.visited())
    gg.close()


def _with_ns(ascii_matrix, rate, seed):
    rng = np.random.default_rng(seed)
    a = ascii_matrix.copy()
    hit = rng.random(a.shape) < rate
    hit[::2] = False
    a[hit] = ord("N")
    return a


@pytest.mark.parametrize("k,mask,G", [
    (40, api.spaced_seed_kmer_pair(40, 12), 60000),
    (33, api.spaced_seed_kmer_pair(33, 11), 40000),
    (64, api.spaced_seed_qr_pair(64, 23), 100000),       # BASELINE.json configs[3] in the small: --qr-seed
    (100, api.spaced_seed_kmer_pair(100, 32), 40000),
])
def test_spaced_seed_matches_oracle(k, mask, G):
    m1, m2 = synth.make_read_set(G, 35.0, err=0.01, genome_seed=k, read_seed=k + 1)
    asc = _with_ns(synth.codes_to_ascii(np.concatenate([m1, m2])), 0.004, k)
    buf, off = api.matrix_to_seqs(asc)
    counters = 1 << 22
    o = ob.Oracle(k, counters=counters, mask=mask.encode())
    g = api.BloomDBG(k, counters=counters, spaced_seed=mask)
    o.load(buf, off)
    g.load(buf, off)
    assert np.array_equal(o.counters(), g.counters())
    po, ho = o.hash_seq(bytes(asc[1]))
    pg, hg = g.hash_seq(bytes(asc[1]))
    assert np.array_equal(po, pg) and np.array_equal(ho, hg)
    ro, co = o.assemble(buf, off)
    rg, cg = g.assemble(buf, off)
    assert np.array_equal(ro, rg)
    assert any(b"N" in c.seq for c in co)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    assert np.array_equal(o.visited(), g.visited())
    assert o.assembly_counters() == g.assembly_counters()


def test_small_claim_table_and_batches_still_exact():
    # tiny claim table / insert batches force many retry rounds; results must not change
    m1, m2 = synth.make_read_set(30000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    o = ob.Oracle(40, counters=1 << 20)
    g = api.BloomDBG(40, counters=1 << 20, insert_batch_kmers=20000, claim_log2=12)
    o.load(buf, off)
    g.load(buf, off)
    assert np.array_equal(o.counters(), g.counters())
    assert g.stats()["insert_rounds"] > 20


@pytest.mark.parametrize("counters", [1 << 22, 1 << 20, 1 << 19])
def test_kmers_raising_shared_counters_settled_by_the_tiles_leave_the_oracles_counters(counters, monkeypatch):
    """PASS 1's round-5 rule on the device (op_verdict / FCoSettle / FCoFinal, tile_apply's repeated raises; the candidates packed per
    wavefront by a ballot): the counter array is the oracle's sequential incrementMin (CountingBloomFilter.hpp:135-162) with the
    rule on, off, cut short after one and two passes (every candidate left then takes the rounds) and with a 1,024-bit table of
    marked counters, at three occupancies; with the rule on an order of magnitude fewer ops take the reservation rounds."""
    m1, m2 = synth.make_read_set(60000, 40.0, err=0.006, genome_seed=11, read_seed=12)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    o = ob.Oracle(48, counters=counters)
    o.load(buf, off)
    want = o.counters()
    pending = {}
    for name, env in (("off", {"ABG_COSETTLE": "0"}), ("on", {}), ("one_pass", {"ABG_COSETTLE_PASSES": "1"}),
                      ("two_passes", {"ABG_COSETTLE_PASSES": "2"}), ("tiny_table", {"ABG_COSETTLE_LOG2": "10"})):
        for key in ("ABG_COSETTLE", "ABG_COSETTLE_PASSES", "ABG_COSETTLE_LOG2"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        g = api.BloomDBG(48, counters=counters, insert_batch_kmers=1 << 17)
        g.load(buf, off)
        st = g.stats()
        assert st["tiled_ops"] > 0 and st["tile_overflows"] == 0, (name, st)
        assert np.array_equal(want, g.counters()), name
        pending[name] = st["tiled_pending"]
        g.close()
    assert pending["on"] * 8 < pending["off"], pending
    assert pending["on"] <= pending["two_passes"] <= pending["one_pass"] <= pending["off"], pending
    assert pending["on"] <= pending["tiny_table"] <= pending["off"], pending


def test_saturating_counters_and_duplicate_kmers():
    reads = [b"ACGTTGCATGCCGATAGCTAGGATCCATGCAAATTTGGCC"] * 300 + [b"A" * 60, b"T" * 60, b"ACAC" * 20]
    buf, off = api.concat_seqs(reads)
    o = ob.Oracle(21, counters=4096)
    g = api.BloomDBG(21, counters=4096)
    o.load(buf, off)
    g.load(buf, off)
    a, b = o.counters(), g.counters()
    assert a.max() == 255
    assert np.array_equal(a, b)
    ro, co = o.assemble(buf, off)
    rg, cg = g.assemble(buf, off)
    assert np.array_equal(ro, rg)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]


def test_a_kmer_that_recurs_thousands_of_times_in_a_batch_does_not_serialise_it(monkeypatch):
    """A homopolymer run, two satellites and the homopolymer's reverse complement at several hundred copies among ordinary reads:
    their k-mers' pairs run the bins of their counters over (no ABG_TILE_CAP here).  The batch is judged and applied through a sort
    of its pairs (Engine::sorted_judge): the sequential filter's counters (CountingBloomFilter.hpp:135-162), and no more than three
    times the reservation rounds the same reads take without the repeats; switched off, the whole batch takes the rounds -- exact
    too, one round per copy."""
    k = 40
    m1, m2 = synth.make_read_set(30000, 30.0)
    plain = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    hot = [b"A" * 150] * 400 + [b"AC" * 75] * 300 + [b"CA" * 75] * 100 + [(b"GATTA" * 30)[i % 5:][:140] for i in range(300)] + [b"T" * 150] * 100
    mixed = plain + hot
    order = np.random.default_rng(4).permutation(len(mixed))
    buf, off = api.concat_seqs([mixed[i] for i in order])
    pbuf, poff = api.concat_seqs(plain)
    counters = 1 << 21
    base = api.BloomDBG(k, counters=counters)
    base.load(pbuf, poff)
    base_rounds = base.stats()["insert_rounds"]
    base.close()
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    g = api.BloomDBG(k, counters=counters)
    g.load(buf, off)
    st = g.stats()
    assert st["tile_overflows"] > 0 and st["tiled_ops"] > 0, st
    assert o.counters().max() == 255 and np.array_equal(o.counters(), g.counters())
    assert st["insert_rounds"] <= 3 * max(base_rounds, 8), (st["insert_rounds"], base_rounds)
    ro, co = o.assemble(buf, off)
    rg, cg = g.assemble(buf, off)
    assert np.array_equal(ro, rg) and [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    g.close()
    monkeypatch.setenv("ABG_SORTED_OVERFLOW", "0")
    old = api.BloomDBG(k, counters=counters)
    old.load(buf, off)
    assert np.array_equal(o.counters(), old.counters())
    assert old.stats()["insert_rounds"] > 20 * st["insert_rounds"]
    old.close()


def test_empty_short_and_non_acgt_inputs():
    g = api.BloomDBG(31, counters=4096)
    buf, off = api.concat_seqs([])
    g.load(buf, off)
    res, contigs = g.assemble(buf, off)
    assert len(res) == 0 and contigs == []
    buf, off = api.concat_seqs([b"ACGT", b"", b"ACGTNNNN" * 10])
    g.load(buf, off)
    assert g.counters().sum() == 0
    res, contigs = g.assemble(buf, off)
    assert list(res) == [1, 1, 2] and contigs == []


def test_chunked_calls_equal_single_call():
    gc = GoldenCase("k32")
    g = api.BloomDBG(**gc.kwargs())
    cut = [0, 1, 777, 2000, gc.n]
    for a, b in zip(cut, cut[1:]):
        g.load(gc.buf[int(gc.off[a]):int(gc.off[b])], gc.off[a:b + 1] - gc.off[a])
    contigs_all, results_all = [], []
    for a, b in zip(cut, cut[1:]):
        r, c = g.assemble(gc.buf[int(gc.off[a]):int(gc.off[b])], gc.off[a:b + 1] - gc.off[a])
        for x in c:
            x.read_index += a
        contigs_all += c
        results_all.append(r)
    assert api.format_fasta(contigs_all, gc.ids) == gc.fasta
    assert api.format_read_log(np.concatenate(results_all), gc.ids) == gc.readlog


def test_read_set_in_several_buffers_is_one_pass():
    """abg_assemble_seqs_v: the reads in four buffers, one pass -- read indices count through them."""
    gc = GoldenCase("k40_mixed")
    g = api.BloomDBG(**gc.kwargs())
    g.load(gc.buf, gc.off)
    cut = [0, 1, 777, 2000, gc.n]
    chunks = [(bytes(gc.buf[int(gc.off[a]):int(gc.off[b])]), gc.off[a:b + 1] - gc.off[a]) for a, b in zip(cut, cut[1:])]
    results, contigs = g.assemble_chunks(chunks)
    assert api.format_fasta(contigs, gc.ids) == gc.fasta
    assert api.format_read_log(results, gc.ids) == gc.readlog
    assert api.format_trace(contigs, gc.ids, gc.reads, gc.opts["k"], with_length=False) == gc.trace


@pytest.mark.parametrize("name", ["k40_mixed", "k64", "k48_K16"])
def test_reads_kept_on_the_device_between_the_passes(name):
    """abg_keep_reads / abg_load_seqs_v / abg_assemble_kept: the reads go in once, in several buffers and
    calls (the device's share of a call running beside the next call's packing), and PASS 2 takes them
    from where PASS 1 put them -- the reference's outputs, verdicts of the rejected reads included."""
    gc = GoldenCase(name)
    g = api.BloomDBG(**gc.kwargs(), spaced_seed=mask_of(gc))
    g.keep_reads(True, len(gc.buf))
    cut = [0, 0, 1, 777, 2000, gc.n]
    chunks = [(bytes(gc.buf[int(gc.off[a]):int(gc.off[b])]), gc.off[a:b + 1] - gc.off[a]) for a, b in zip(cut, cut[1:])]
    g.load_chunks(chunks[:2])
    g.load_chunks(chunks[2:3])
    g.load(*chunks[3])
    g.load_chunks(chunks[4:])
    assert g.counting_stats()[1] == gc.meta["filtered_popcount"]
    results, contigs = g.assemble_kept(gc.n)
    assert api.format_fasta(contigs, gc.ids) == gc.fasta
    assert api.format_read_log(results, gc.ids) == gc.readlog
    assert api.format_trace(contigs, gc.ids, gc.reads, gc.opts["k"], with_length=False) == gc.trace
    with pytest.raises(api.AbyssAmdError):
        g.assemble_kept(0)  # nothing is kept any more
    # and a store the device has no room for is refused up front
    with pytest.raises(api.AbyssAmdError):
        g.keep_reads(True, 1 << 42)


def test_export_import_roundtrip_and_idempotence():
    # size-independent properties on a larger set: (1) filters survive export/import into a
    # fresh context and give the same assembly (the -i prebuilt path, bloom-dbg.cc:302-343);
    # (2) assembling the same reads again yields no new contig and every solid read is visited
    m1, m2 = synth.make_read_set(400000, 40.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    g = api.BloomDBG(64, counters=1 << 26)
    g.load(buf, off)
    cnt = g.counters()
    r1, c1 = g.assemble(buf, off)
    assert sum(not c.redundant for c in c1) > 50
    r2, c2 = g.assemble(buf, off)
    assert all(c.redundant for c in c2) or c2 == []
    solid = np.isin(r1, [5, 7])
    assert np.all(np.isin(r2[solid], [5, 7]))
    assert np.array_equal(r1 == 3, r2 == 3) and np.array_equal(r1 == 4, r2 == 4)
    h = api.BloomDBG(64, counters=1 << 26)
    h.set_counters_array(cnt)
    r3, c3 = h.assemble(buf, off)
    assert np.array_equal(r1, r3)
    assert [contig_tuple(c) for c in c1] == [contig_tuple(c) for c in c3]
    # every output unitig consists of solid k-mers only: re-loading the unitigs' k-mers into
    # the assembled filter of a third context leaves all reads that were visited, visited
    assert np.array_equal(g.counters(), cnt)


@pytest.mark.parametrize("name", ["k64", "k40_mixed", "k48_K16"])
def test_graphviz_dump_matches_reference(name):
    """-g (abg_output_graph_seqs): trimSeq + the breadth-first searches on the device against the file
    the unmodified reference wrote for the same reads (tests/golden/make_graph_golden.py)."""
    import hashlib
    import json
    import os
    from util import GOLDEN, mask_of
    g0 = GoldenCase(name)
    kw = g0.kwargs()
    g = api.BloomDBG(kw["k"], counters=g0.meta["counters"], num_hashes=kw["num_hashes"], min_cov=kw["min_cov"],
                     trim=kw["trim"], spaced_seed=mask_of(g0), claim_log2=22)
    g.load(g0.buf, g0.off)
    text, nodes, edges = g.output_graph(g0.buf, g0.off)
    ref = json.load(open(os.path.join(GOLDEN, "graph_golden.json")))[name]
    assert (len(text), nodes, edges) == (ref["bytes"], ref["nodes"], ref["edges"])
    assert hashlib.sha256(text).hexdigest() == ref["sha256"]
    again, n2, e2 = g.output_graph(g0.buf, g0.off, frame=False)
    assert again == b"" and (n2, e2) == (0, 0)
    g.close()


@pytest.mark.parametrize("env", [{"ABG_TILED": "0"}, {"ABG_TILE_CAP": "300"}, {"ABG_GUIDE_STRIDE": "0", "ABG_MEMO": "0"},
                                 {"ABG_GUIDE_STRIDE": "1"}, {"ABG_GUIDE_SEEN": "0"}, {"ABG_PAR_COMMIT_MAX_GB": "0", "ABG_T_TAGS": "4"},
                                 {"ABG_COMPACT_THRESHOLD": "1"}, {"ABG_OVERLAP_BINS": "0"}, {"ABG_OVERLAP_BINS": "1", "ABG_TILE_CAP": "300"},
                                 {"ABG_P2_FIRST_BATCH": "256"}, {"ABG_P2_FIRST_BATCH": "256", "ABG_CLS_ARCHIVE": "0"},
                                 {"ABG_P2_FIRST_BATCH": "256", "ABG_CLS_ARCHIVE_MAX_MB": "0"}])
def test_accelerators_and_fallbacks_do_not_change_results(env, monkeypatch):
    """Every accelerator has an exact slow path behind it and every table a fallback: PASS 1 without the
    LDS tiles / with bins that overflow, walkers without guide and memo / with the densest guide, the
    commit with hashed time stamps, flagged-and-compacted losers in every round,
    PASS 1 without the next batch staged on the side stream / with a staged batch whose bins overflow, the classification with
    the archive of committed contigs over many small batches / without it / with an archive too small for the contigs (what does
    not fit is not archived).  Same bytes as the reference, and the work counters show the path was taken."""
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    gc = GoldenCase("k64")
    g = api.BloomDBG(insert_batch_kmers=1 << 17, **gc.kwargs())
    g.load(gc.buf, gc.off)
    assert g.counting_stats()[1] == gc.meta["filtered_popcount"]
    results, contigs = g.assemble(gc.buf, gc.off)
    st = g.stats()
    if env.get("ABG_TILED") == "0":
        assert st["tiled_ops"] == 0
    elif "ABG_TILE_CAP" in env:
        assert st["tile_overflows"] > 0
    else:
        assert st["tiled_ops"] > 0 and st["tile_overflows"] == 0
    if env.get("ABG_GUIDE_STRIDE") == "0":
        assert st["bulk_steps"] == 0 and st["memo_hits"] == 0
    else:
        assert st["bulk_steps"] > st["lin_steps"] and st["memo_hits"] > 0
    if "ABG_P2_FIRST_BATCH" in env:
        if env.get("ABG_CLS_ARCHIVE") == "0":
            assert st["cls_covered_reads"] == 0 and st["archive_bases"] == 0, st
        elif "ABG_CLS_ARCHIVE_MAX_MB" in env:
            assert 0 < st["archive_bases"] < 20000 and st["walk_rounds"] > 3, st
        else:
            assert st["cls_covered_reads"] > 100 and st["archive_bases"] > 15000 and st["walk_rounds"] > 3, st
    assert api.format_fasta(contigs, gc.ids) == gc.fasta
    assert api.format_read_log(results, gc.ids) == gc.readlog
    assert api.format_trace(contigs, gc.ids, gc.reads, gc.opts["k"], with_length=False) == gc.trace
    g.close()
