"""The product's device logic (abyss_amd/csrc/*.h) executed serially on the CPU through
tests/hostcheck, against the oracle and the golden fixtures.  This checks the kernels'
LOGIC without a GPU; the kernels themselves are checked by the -m gpu tests."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import _lib, api, build, synth
from util import GOLDEN, GoldenCase, contig_tuple, mask_of, random_reads


class HostCheck:
    def __init__(self, k, counters, num_hashes=4, min_cov=2, trim=None, insert_batch=0, claim_log2=0, p2_first=0, mask=None):
        l = C.CDLL(os.environ.get("ABG_HOSTCHECK_LIB") or build.build_hostcheck())
        l.hc_create.restype = C.c_void_p
        l.hc_create.argtypes = [C.c_uint] * 4 + [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64, C.c_char_p]
        l.hc_destroy.argtypes = [C.c_void_p]
        l.hc_size.restype = C.c_uint64
        l.hc_size.argtypes = [C.c_void_p]
        l.hc_counters.restype = C.POINTER(C.c_uint8)
        l.hc_counters.argtypes = [C.c_void_p]
        l.hc_visited.restype = C.POINTER(C.c_uint8)
        l.hc_visited.argtypes = [C.c_void_p]
        l.hc_load_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        l.hc_popcounts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        l.hc_assemble_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, _lib.CONTIG_CB, C.c_void_p]
        l.hc_hash_seq.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        l.hc_get_counters.argtypes = [C.c_void_p, C.POINTER(_lib.Counters)]
        l.hc_get_stats.argtypes = [C.c_void_p, C.POINTER(_lib.Stats)]
        l.hc_mod_check.restype = C.c_uint64
        l.hc_mod_check.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
        self.l = l
        self.k, self.num_hashes = k, num_hashes
        self.h = l.hc_create(k, num_hashes, min_cov, k if trim is None else trim, counters, insert_batch, claim_log2, p2_first,
                               mask.encode() if mask else None)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.l.hc_destroy(self.h)

    @property
    def size(self):
        return self.l.hc_size(self.h)

    def load(self, buf, off):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        assert self.l.hc_load_seqs(self.h, buf, off.ctypes.data, len(off) - 1) == 0

    def counters(self):
        # (through the export: on a sliced filter -- tests/test_dist_partition.py -- the ranks pass their ranges around)
        out = np.empty(self.size, dtype=np.uint8)
        self.l.hc_counters_export.argtypes = [C.c_void_p, C.c_void_p]
        self.l.hc_last_error.restype = C.c_char_p
        self.l.hc_last_error.argtypes = [C.c_void_p]
        rc = self.l.hc_counters_export(self.h, out.ctypes.data)
        assert rc == 0, (rc, self.l.hc_last_error(self.h))
        return out

    def visited(self):
        return np.ctypeslib.as_array(self.l.hc_visited(self.h), (self.size // 8,)).copy()

    def counting_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.l.hc_popcounts(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def assemble(self, buf, off):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        res = np.zeros(n, dtype=np.uint8)
        out = []

        def cb(_u, c):
            c = c.contents
            out.append(api.ContigRecord(c.contig_id, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                        c.right_ext, c.left_code, c.right_code, c.seed_pos))
        assert self.l.hc_assemble_seqs(self.h, buf, off.ctypes.data, n, res.ctypes.data, _lib.CONTIG_CB(cb), None) == 0
        return res, out

    def assemble_chunks(self, chunks):
        """abg_assemble_seqs_v: several (buf, off) chunks, one pass; read indices count through them."""
        import ctypes as C
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for _, o in chunks]
        nc = len(chunks)
        seqs_v = (C.c_char_p * nc)(*[b for b, _ in chunks])
        off_v = (C.c_void_p * nc)(*[o.ctypes.data for o in offs])
        n_v = (C.c_uint64 * nc)(*[len(o) - 1 for o in offs])
        res = np.zeros(sum(len(o) - 1 for o in offs), dtype=np.uint8)
        out = []

        def cb(_u, c):
            c = c.contents
            out.append(api.ContigRecord(c.contig_id, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                        c.right_ext, c.left_code, c.right_code, c.seed_pos))
        self.l.hc_assemble_seqs_v.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _lib.CONTIG_CB, C.c_void_p]
        assert self.l.hc_assemble_seqs_v(self.h, nc, C.cast(seqs_v, C.c_void_p), C.cast(off_v, C.c_void_p), C.cast(n_v, C.c_void_p),
                                         res.ctypes.data, _lib.CONTIG_CB(cb), None) == 0
        return res, out

    def keep_reads(self, on=True, expected_bases=0):
        self.l.hc_keep_reads.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        return self.l.hc_keep_reads(self.h, int(on), expected_bases)

    def load_chunks(self, chunks):
        """abg_load_seqs_v: several (buf, off) chunks, one call."""
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for _, o in chunks]
        nc = len(chunks)
        seqs_v = (C.c_char_p * nc)(*[b for b, _ in chunks])
        off_v = (C.c_void_p * nc)(*[o.ctypes.data for o in offs])
        n_v = (C.c_uint64 * nc)(*[len(o) - 1 for o in offs])
        self.l.hc_load_seqs_v.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        assert self.l.hc_load_seqs_v(self.h, nc, C.cast(seqs_v, C.c_void_p), C.cast(off_v, C.c_void_p), C.cast(n_v, C.c_void_p)) == 0

    def assemble_kept(self, n):
        res = np.zeros(max(n, 1), dtype=np.uint8)
        out = []

        def cb(_u, c):
            c = c.contents
            out.append(api.ContigRecord(c.contig_id, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                        c.right_ext, c.left_code, c.right_code, c.seed_pos))
        self.l.hc_assemble_kept.argtypes = [C.c_void_p, C.c_void_p, _lib.CONTIG_CB, C.c_void_p]
        rc = self.l.hc_assemble_kept(self.h, res.ctypes.data, _lib.CONTIG_CB(cb), None)
        return rc, res[:n], out

    def hash_seq(self, seq):
        cap = max(len(seq), 1)
        pos = np.zeros(cap, dtype=np.uint32)
        hashes = np.zeros((cap, self.num_hashes), dtype=np.uint64)
        n = C.c_uint64()
        assert self.l.hc_hash_seq(self.h, seq, len(seq), pos.ctypes.data, hashes.ctypes.data, cap, C.byref(n)) == 0
        return pos[:n.value], hashes[:n.value]

    def output_graph(self, buf, off, frame=True):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        parts = [b"digraph g {\n"] if frame else []
        cb = _lib.TEXT_CB(lambda _u, p, n: parts.append(C.string_at(p, n)))
        a, b = C.c_uint64(), C.c_uint64()
        self.l.hc_output_graph_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, _lib.TEXT_CB, C.c_void_p,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        assert self.l.hc_output_graph_seqs(self.h, buf, off.ctypes.data, len(off) - 1, cb, None, C.byref(a), C.byref(b)) == 0
        if frame:
            parts.append(b"}\n")
        return b"".join(parts), a.value, b.value

    def assembly_counters(self):
        c = _lib.Counters()
        self.l.hc_get_counters(self.h, C.byref(c))
        return {n: getattr(c, n) for n, _ in c._fields_}

    def stats(self):
        s = _lib.Stats()
        self.l.hc_get_stats(self.h, C.byref(s))
        return {n: getattr(s, n) for n, _ in s._fields_}


def test_exact_modulo():
    hc = HostCheck(8, 1024)
    rng = np.random.default_rng(0)
    edge = np.array([0, 1, 2, 2 ** 32 - 1, 2 ** 32, 2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, 2 ** 64 - 2], dtype=np.uint64)
    for m in (1, 2, 3, 64, 1000, 3728320, 93206784, 1908874368, 38177487104, 477218588480, 2 ** 40, 2 ** 63 + 12345,
              2 ** 64 - 59):
        hs = np.concatenate([edge, rng.integers(0, 2 ** 64, size=200000, dtype=np.uint64),
                             np.uint64(m) * rng.integers(0, max(1, (2 ** 64 - 1) // m), size=1000, dtype=np.uint64),
                             np.uint64(m) * rng.integers(1, max(2, (2 ** 64 - 1) // m), size=1000, dtype=np.uint64) - np.uint64(1)])
        assert hc.l.hc_mod_check(m, hs.ctypes.data, len(hs)) == 0, m


def test_hash_stream_matches_reference_vectors():
    vectors = json.load(open(os.path.join(GOLDEN, "nthash_vectors.json")))
    for v in vectors:
        hc = HostCheck(v["k"], 1024)
        pos, h = hc.hash_seq(v["seq"].encode())
        assert list(pos) == v["pos"]
        assert [[str(int(x)) for x in row] for row in h] == v["hashes"]


@pytest.mark.parametrize("name", ["k32", "k64", "k25_h3_kc3_t40", "k40_mixed", "k96", "k48_K16", "k50_qr11"])
def test_device_logic_reproduces_reference_run(name):
    g = GoldenCase(name)
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                   claim_log2=16, p2_first=128, mask=mask_of(g))
    hc.load(g.buf, g.off)
    assert hc.counting_stats()[1] == g.meta["filtered_popcount"]
    results, contigs = hc.assemble(g.buf, g.off)
    assert api.format_fasta(contigs, g.ids) == g.fasta
    assert api.format_read_log(results, g.ids) == g.readlog
    assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace
    c = hc.assembly_counters()
    assert (c["reads_processed"], c["solid_reads"], c["visited_reads"]) == (
        g.meta["reads"], g.meta["solid_reads"], g.meta["visited_reads"])


@pytest.mark.parametrize("name", ["s_plasmids_k32", "s_tandem_k32", "s_tandem_k64_t20", "s_inverted_k40", "s_lowcomplex_k25", "s_plasmids_k48_K16", "s_mixed_k192", "s_mixed_k12", "s_mixed_k32_H1", "s_mixed_k40_H6", "s_mixed_k32_H12_kc3", "s_satellite_k40"])
def test_device_logic_reproduces_reference_run_on_cycles_repeats_and_hairpins(name):
    """Graph shapes a random linear genome never makes (tests/golden/make_structured.py, from the unmodified reference at -j1):
    circular replicons, tandem repeats with units shorter and longer than k, inverted repeats and hairpins, homopolymer and
    dinucleotide runs -- what Unittest/Graph/ExtendPathTest.cpp's cycles / cyclesAndBranches / longestBranch / withTrimming
    cases are about, as k-mer graphs."""
    test_device_logic_reproduces_reference_run(name)


@pytest.mark.parametrize("k,G", [(17, 15000), (33, 15000), (65, 15000), (97, 20000), (129, 15000), (150, 12000)])
def test_device_logic_matches_oracle(k, G):
    m1, m2 = synth.make_read_set(G, 25.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    counters = 1 << 21
    o = ob.Oracle(k, counters=counters)
    hc = HostCheck(k, counters, insert_batch=20000, claim_log2=14, p2_first=64)
    o.load(buf, off)
    hc.load(buf, off)
    assert np.array_equal(o.counters(), hc.counters())
    assert o.counting_stats() == hc.counting_stats()
    ro, co = o.assemble(buf, off)
    rh, ch = hc.assemble(buf, off)
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert np.array_equal(o.visited(), hc.visited())
    assert o.assembly_counters() == hc.assembly_counters()


def _with_ns(ascii_matrix, rate, seed):
    """Replace a fraction `rate` of the characters by 'N' (in half of the reads only, so that PASS 2
    still sees plenty of pure-ACGT reads)."""
    rng = np.random.default_rng(seed)
    a = ascii_matrix.copy()
    hit = rng.random(a.shape) < rate
    hit[::2] = False
    a[hit] = ord("N")
    return a


@pytest.mark.parametrize("k,mask,G", [
    (40, api.spaced_seed_kmer_pair(40, 12), 15000),
    (33, api.spaced_seed_kmer_pair(33, 11), 12000),          # odd k: the isCanonical tie quirk under a mask
    (64, api.spaced_seed_qr_pair(64, 23), 15000),
    (31, "1101101011110101010111101011011", 12000),          # -s with a dense pattern (symmetric)
    (100, api.spaced_seed_kmer_pair(100, 32), 15000),
])
def test_spaced_seed_device_logic_matches_oracle(k, mask, G):
    """Spaced seeds (-K / --qr-seed / -s): reads with 'N' under a '0' still yield k-mers in PASS 1
    (RollingHashIterator.h:35-97), contigs shorter than 2k-1 can carry 'N' columns (pathToSeq)."""
    assert mask == mask[::-1] and len(mask) == k
    m1, m2 = synth.make_read_set(G, 35.0, err=0.01, genome_seed=k, read_seed=k + 1)
    asc = _with_ns(synth.codes_to_ascii(np.concatenate([m1, m2])), 0.004, k)
    buf, off = api.matrix_to_seqs(asc)
    counters = 1 << 21
    o = ob.Oracle(k, counters=counters, mask=mask.encode())
    hc = HostCheck(k, counters, insert_batch=20000, claim_log2=14, p2_first=64, mask=mask)
    o.load(buf, off)
    hc.load(buf, off)
    assert np.array_equal(o.counters(), hc.counters())
    seq = bytes(asc[1])
    po, ho = o.hash_seq(seq)
    ph, hh = hc.hash_seq(seq)
    assert np.array_equal(po, ph) and np.array_equal(ho, hh)
    ro, co = o.assemble(buf, off)
    rh, ch = hc.assemble(buf, off)
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert np.array_equal(o.visited(), hc.visited())
    assert o.assembly_counters() == hc.assembly_counters()


def test_parallel_commit_needs_several_passes_and_stays_exact():
    """A filter so small that false positives chain reads and contigs together: the fixed-point
    commit needs more than one pass per batch here, and must still equal the sequential order."""
    k, counters = 25, 1 << 17
    m1, m2 = synth.make_read_set(8000, 30.0, err=0.02, genome_seed=30, read_seed=34)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    o = ob.Oracle(k, counters=counters)
    hc = HostCheck(k, counters, insert_batch=20000, claim_log2=14, p2_first=64)
    o.load(buf, off)
    hc.load(buf, off)
    ro, co = o.assemble(buf, off)
    rh, ch = hc.assemble(buf, off)
    st = hc.stats()
    assert st["commit_rounds"] > st["walk_rounds"]
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert np.array_equal(o.visited(), hc.visited())
    assert o.assembly_counters() == hc.assembly_counters()


def test_time_stamp_tags_wrap_around(monkeypatch):
    """The parallel commit's time stamps are cleared only when the pass tags run out (1024 passes);
    ABG_T_TAGS=3 makes that happen every other pass."""
    monkeypatch.setenv("ABG_T_TAGS", "3")
    for name in ("k32", "k48_K16"):
        g = GoldenCase(name)
        kw = g.kwargs()
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=64, mask=mask_of(g))
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        assert hc.stats()["commit_rounds"] >= 6
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog


def test_ordered_commit_kernel_still_exact(monkeypatch):
    """ABG_PAR_COMMIT=0 selects the single-workgroup ordered commit (the fallback when the time
    stamps of the parallel form do not fit in memory)."""
    monkeypatch.setenv("ABG_PAR_COMMIT", "0")
    g = GoldenCase("k40_mixed")
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                   claim_log2=16, p2_first=128)
    hc.load(g.buf, g.off)
    results, contigs = hc.assemble(g.buf, g.off)
    assert hc.stats()["commit_rounds"] == 0
    assert api.format_fasta(contigs, g.ids) == g.fasta
    assert api.format_read_log(results, g.ids) == g.readlog
    assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_contains_seq_is_min_count_at_least_kc():
    """abg_contains_seq (the -C / -R coverage track): goodKmerSet.contains() per valid k-mer."""
    g = GoldenCase("k40_mixed")
    kw = g.kwargs()
    o = ob.Oracle(kw["k"], counters=g.meta["counters"], num_hashes=kw["num_hashes"], min_cov=kw["min_cov"])
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], insert_batch=50000, claim_log2=16)
    o.load(g.buf, g.off)
    hc.load(g.buf, g.off)
    hc.l.hc_contains_seq.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    ref_like = b"".join(g.reads[:40]) + b"NNacgtn" + g.reads[50] + b"ACGTACGTAC" * 30
    pos = np.zeros(len(ref_like), dtype=np.uint32)
    val = np.zeros(len(ref_like), dtype=np.uint8)
    n = C.c_uint64()
    assert hc.l.hc_contains_seq(hc.h, ref_like, len(ref_like), pos.ctypes.data, val.ctypes.data, len(ref_like), C.byref(n)) == 0
    po, ho = o.hash_seq(ref_like)
    assert np.array_equal(po, pos[:n.value])
    assert np.array_equal(o.min_count(ho) >= kw["min_cov"], val[:n.value].astype(bool))
    assert 0 < val[:n.value].sum() < n.value


def test_reset_gives_a_fresh_context():
    g = GoldenCase("k32")
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], insert_batch=30000, claim_log2=16, p2_first=100)
    for _ in range(2):
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog
        hc.l.hc_reset.argtypes = [C.c_void_p]
        hc.l.hc_reset(hc.h)
        assert hc.counters().sum() == 0 and hc.visited().sum() == 0
        assert hc.assembly_counters()["next_contig_id"] == 0


def test_saturating_counters_and_duplicate_kmers():
    # heavy duplication: 300 copies of the same read saturate counters at 255
    # (CountingBloomFilter.hpp:146-149); homopolymers give runs of identical k-mers
    reads = [b"ACGTTGCATGCCGATAGCTAGGATCCATGCAAATTTGGCC"] * 300 + [b"A" * 60, b"T" * 60, b"ACAC" * 20]
    buf, off = api.concat_seqs(reads)
    o = ob.Oracle(21, counters=4096)
    hc = HostCheck(21, 4096, insert_batch=1000, claim_log2=8)
    o.load(buf, off)
    hc.load(buf, off)
    a, b = o.counters(), hc.counters()
    assert a.max() == 255
    assert np.array_equal(a, b)


def test_empty_and_short_inputs():
    hc = HostCheck(31, 4096)
    buf, off = api.concat_seqs([])
    hc.load(buf, off)
    res, contigs = hc.assemble(buf, off)
    assert len(res) == 0 and contigs == []
    buf, off = api.concat_seqs([b"ACGT", b"", b"ACGTNNNN" * 10])
    hc.load(buf, off)
    assert hc.counters().sum() == 0
    res, contigs = hc.assemble(buf, off)
    assert list(res) == [1, 1, 2] and contigs == []  # SHORTER_THAN_K, SHORTER_THAN_K, NON_ACGT


def test_chunked_calls_equal_single_call():
    g = GoldenCase("k32")
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], insert_batch=30000, claim_log2=16, p2_first=100)
    cut = [0, 1, 777, 2000, g.n]
    for a, b in zip(cut, cut[1:]):
        off = g.off[a:b + 1] - g.off[a]
        hc.load(g.buf[int(g.off[a]):int(g.off[b])], off)
    contigs_all, results_all = [], []
    for a, b in zip(cut, cut[1:]):
        off = g.off[a:b + 1] - g.off[a]
        r, c = hc.assemble(g.buf[int(g.off[a]):int(g.off[b])], off)
        for x in c:
            x.read_index += a
        contigs_all += c
        results_all.append(r)
    assert api.format_fasta(contigs_all, g.ids) == g.fasta
    assert api.format_read_log(np.concatenate(results_all), g.ids) == g.readlog


@pytest.mark.parametrize("name", ["k32", "k40_mixed", "k48_K16", "k25_h3_kc3_t40"])
def test_graphviz_dump_device_logic_matches_reference(name):
    """-g: trimSeq + the breadth-first searches (FTrimRun, FGraphBfs) and the host's replay against the
    file the unmodified reference wrote (SHA-256, size, visitor counters); the vertex table and the node
    buffer start small, so both grow on the way."""
    import hashlib
    g = GoldenCase(name)
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                   claim_log2=16, mask=mask_of(g))
    hc.load(g.buf, g.off)
    text, nodes, edges = hc.output_graph(g.buf, g.off)
    ref = json.load(open(os.path.join(GOLDEN, "graph_golden.json")))[name]
    assert (len(text), nodes, edges) == (ref["bytes"], ref["nodes"], ref["edges"])
    assert hashlib.sha256(text).hexdigest() == ref["sha256"]


def test_graphviz_dump_in_chunks_equals_single_call():
    """The visited-vertex set carries over between calls like the reference's colour map over files."""
    g = GoldenCase("k32")
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], insert_batch=50000, claim_log2=16)
    hc.load(g.buf, g.off)
    whole, n1, e1 = hc.output_graph(g.buf, g.off, frame=False)
    hc2 = HostCheck(kw["k"], g.meta["counters"], insert_batch=50000, claim_log2=16)
    hc2.load(g.buf, g.off)
    parts, nn, ee = [], 0, 0
    cut = [0, 1, 50, 51, 1000, g.n]
    for a, b in zip(cut, cut[1:]):
        t, n, e = hc2.output_graph(g.buf[int(g.off[a]):int(g.off[b])], g.off[a:b + 1] - g.off[a], frame=False)
        parts.append(t); nn += n; ee += e
    assert b"".join(parts) == whole and (nn, ee) == (n1, e1)
    # nothing is printed twice: a second pass over the same reads finds every start vertex black
    again, n2, e2 = hc2.output_graph(g.buf, g.off, frame=False)
    assert again == b"" and (n2, e2) == (0, 0)


@pytest.mark.parametrize("name,split", [("k40_mixed", 7), ("k48_K16", 3), ("k25_h3_kc3_t40", 16)])
def test_host_packing_on_several_threads_is_invisible(name, split, monkeypatch):
    """The host packs reads into the 2-bit device layout on several threads (ranges of reads, joined in
    order); ABG_HOST_SPLIT forces that on inputs this small.  Reads with Ns, lower case and short reads
    included (k40_mixed)."""
    monkeypatch.setenv("ABG_HOST_SPLIT", str(split))
    g = GoldenCase(name)
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                   claim_log2=16, p2_first=128, mask=mask_of(g))
    hc.load(g.buf, g.off)
    assert hc.counting_stats()[1] == g.meta["filtered_popcount"]
    results, contigs = hc.assemble(g.buf, g.off)
    assert api.format_fasta(contigs, g.ids) == g.fasta
    assert api.format_read_log(results, g.ids) == g.readlog
    assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_hashed_time_stamps_stay_exact(monkeypatch):
    """ABG_PAR_COMMIT_MAX_GB=0: the parallel commit keeps its time stamps in a hash table keyed by bit
    position (what a filter too large for one stamp per bit gets, e.g. B=40G) instead of one per bit."""
    monkeypatch.setenv("ABG_PAR_COMMIT_MAX_GB", "0")
    monkeypatch.setenv("ABG_T_TAGS", "5")
    for name in ("k64", "k48_K16"):
        g = GoldenCase(name)
        kw = g.kwargs()
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=64, mask=mask_of(g))
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        assert hc.stats()["commit_rounds"] >= 6
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog
        assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_device_sized_fast_memory_runs_the_chain_searches(monkeypatch):
    """HC_FAST_BYTES=16384: the walkers get what LDS gives them on the device, so successor()'s chain
    searches (chain_true_branches, chain_bulk) and the bulk scratch run as they do there."""
    monkeypatch.setenv("HC_FAST_BYTES", "20480")
    for name in ("k64", "k96", "k25_h3_kc3_t40", "k48_K16", "k50_qr11"):
        g = GoldenCase(name)
        kw = g.kwargs()
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=128, mask=mask_of(g))
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        st = hc.stats()
        # (odd k and spaced seeds included: the identity of a predicted vertex is vtx_ident's)
        assert st["bulk_steps"] > 5 * st["lin_steps"] and st["chain_steps"] > 0, (name, st)
        assert st["memo_hits"] > 0
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog
        assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_guide_and_memo_do_not_change_results(monkeypatch):
    """The guide table and the successor() memo are accelerators: with both off (ABG_GUIDE_STRIDE=0,
    ABG_MEMO=0), with the densest guide (stride 1), and with that guide but nothing kept of what the bulk steps find out about
    a read's k-mers (ABG_GUIDE_SEEN=0: every walker probes again) the outputs are the reference's."""
    g = GoldenCase("k64")
    kw = g.kwargs()
    for env in ({"ABG_GUIDE_STRIDE": "0", "ABG_MEMO": "0"}, {"ABG_GUIDE_STRIDE": "1"}, {"ABG_GUIDE_STRIDE": "1", "ABG_GUIDE_SEEN": "0"}):
        for key in ("ABG_GUIDE_STRIDE", "ABG_MEMO", "ABG_GUIDE_SEEN"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=128)
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        st = hc.stats()
        if env.get("ABG_GUIDE_STRIDE") == "0":
            assert st["bulk_steps"] == 0 and st["memo_hits"] == 0
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog
        assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


@pytest.mark.parametrize("name", ["k64", "k40_mixed", "s_tandem_k32", "s_inverted_k40", "s_lowcomplex_k25", "s_satellite_k40", "k48_K16"])
def test_archive_of_committed_contigs_does_not_change_verdicts(name, monkeypatch):
    """Round 6: the classification takes the k-mers of a read that lie on a contig an earlier batch committed from an archive of those
    contigs (ContigArchive: a byte per base, a k-mer -> position table filled by FPcApply) instead of probing the two filters for
    each (bloom-dbg.h:58-77,816-828).  Many small batches, so that most reads meet the archive: the reference's read log (every
    verdict), FASTA and trace with the archive on, off, and with one far too small for the contigs (what does not fit is not
    archived, the table's slots collide) -- on random genomes and on the shapes where a read matches a contig on either strand,
    on several contigs or on itself (tandem and inverted repeats, homopolymer runs, satellites), and under a spaced seed."""
    g = GoldenCase(name)
    kw = g.kwargs()
    covered = {}
    for tag, env in (("on", {}), ("off", {"ABG_CLS_ARCHIVE": "0"}), ("tiny", {"ABG_CLS_ARCHIVE_MAX_MB": "0"})):
        for key in ("ABG_CLS_ARCHIVE", "ABG_CLS_ARCHIVE_MAX_MB"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000, claim_log2=16,
                       p2_first=64, mask=mask_of(g))
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        st = hc.stats()
        covered[tag] = (st["cls_covered_reads"], st["archive_bases"], st["cls_decided_reads"])
        assert api.format_fasta(contigs, g.ids) == g.fasta, tag
        assert api.format_read_log(results, g.ids) == g.readlog, tag
        assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace, tag
    assert covered["off"] == (0, 0, 0) and covered["on"][0] > 0 and covered["on"][1] >= covered["tiny"][1] > 0, covered
    # ... and reads well inside one archived contig get their whole verdict there, the two blunt-end look-aheads included
    # (arc_ends_decided: the conditions under which lookAhead cannot fail on a contig's own path) -- not under a spaced seed
    if name in ("k64", "k40_mixed"):
        assert covered["on"][2] > 100, covered
    if name == "k48_K16":
        assert covered["on"][2] == 0, covered


def test_packed_reads_get_their_prefix_sums_and_batches_on_the_device():
    """abg_load_packed: the k-mer prefix sums (FKmerCounts + scan) and the batches' op ranges
    (FCutRanges) are made where the reads are.  Ragged lengths, batches of a few hundred ops (many
    ranges, one of them a single sequence longer than a batch would be refused: not here), the counters
    of the oracle; a sequence shorter than k is an error."""
    import ctypes as C
    rng = np.random.default_rng(11)
    k = 31
    reads = [bytes(rng.choice(list(b"ACGT"), size=int(L)).astype(np.uint8)) for L in rng.integers(k, 140, size=700)]
    buf, off = api.concat_seqs(reads)
    o = ob.Oracle(k, counters=1 << 16)
    o.load(buf, off)
    hc = HostCheck(k, 1 << 16, insert_batch=400, claim_log2=12)
    # pack: 16 bases per word, every sequence on a word boundary
    words, woff, lens = [], [0], []
    for r in reads:
        codes = np.array([b"ACGT".index(bytes([c])) for c in r], dtype=np.uint64)
        pad = np.zeros((len(r) + 15) // 16 * 16, dtype=np.uint64)
        pad[:len(r)] = codes
        words.append((pad.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32))
        woff.append(woff[-1] + len(words[-1]))
        lens.append(len(r))
    words = np.ascontiguousarray(np.concatenate(words)); woff = np.array(woff, dtype=np.uint64); lens = np.array(lens, dtype=np.uint32)
    vp = C.c_void_p
    hc.l.hc_load_packed.argtypes = [vp, vp, vp, vp, C.c_uint64]
    assert hc.l.hc_load_packed(hc.h, words.ctypes.data, woff.ctypes.data, lens.ctypes.data, len(reads)) == 0
    assert np.array_equal(o.counters(), hc.counters())
    lens[5] = k - 1
    assert hc.l.hc_load_packed(hc.h, words.ctypes.data, woff.ctypes.data, lens.ctypes.data, len(reads)) != 0


def test_read_set_in_several_chunks_is_one_pass():
    """abg_assemble_seqs_v: the reads of a golden run handed over in three buffers (the second one
    holding reads that are too short or not ACGT as well) give the reference's outputs, with read
    indices counting through the chunks."""
    g = GoldenCase("k40_mixed")
    kw = g.kwargs()
    hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                   claim_log2=16, p2_first=128)
    hc.load(g.buf, g.off)
    n = len(g.off) - 1
    cuts = [0, n // 5, n // 2, n]
    chunks = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(g.off[a]), int(g.off[b])
        chunks.append((bytes(g.buf[lo:hi]), np.asarray(g.off[a:b + 1], dtype=np.uint64) - np.uint64(lo)))
    results, contigs = hc.assemble_chunks(chunks)
    assert api.format_fasta(contigs, g.ids) == g.fasta
    assert api.format_read_log(results, g.ids) == g.readlog
    assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_chunks_that_are_empty_or_hold_only_rejected_reads():
    """abg_assemble_seqs_v with an empty buffer, a buffer of reads that are all too short or not ACGT, and
    real reads in between: the verdicts and contigs of the one-buffer call, indices counting through."""
    g = GoldenCase("k32")
    kw = g.kwargs()
    n = len(g.off) - 1
    junk = [b"ACGT", b"ACGTNNNNNNACGTACGTACGTACGTACGTACGTACGTACGT", b"", b"acgtn"]
    jbuf, joff = api.concat_seqs(junk)

    def cut(a, b):
        lo, hi = int(g.off[a]), int(g.off[b])
        return bytes(g.buf[lo:hi]), np.asarray(g.off[a:b + 1], dtype=np.uint64) - np.uint64(lo)
    chunks = [(b"", np.zeros(1, dtype=np.uint64)), cut(0, n // 3), (jbuf, joff), cut(n // 3, n), (b"", np.zeros(1, dtype=np.uint64))]
    # the same reads in one buffer
    one_buf = cut(0, n // 3)[0] + jbuf + cut(n // 3, n)[0]
    lens = np.concatenate([np.diff(cut(0, n // 3)[1]), np.diff(joff), np.diff(cut(n // 3, n)[1])])
    one_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    outs = []
    for mode in ("chunks", "one"):
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=256)
        hc.load(g.buf, g.off)
        res, contigs = hc.assemble_chunks(chunks) if mode == "chunks" else hc.assemble(one_buf, one_off)
        outs.append((res.tolist(), [(c.contig_id, c.read_index, c.seq, c.coverage, c.redundant) for c in contigs]))
    assert outs[0] == outs[1]
    res = outs[0][0]
    j0 = n // 3
    assert res[j0:j0 + 4] == [1, 2, 1, 1]  # SHORTER_THAN_K, NON_ACGT, SHORTER_THAN_K, SHORTER_THAN_K


def test_walkers_running_out_of_pool_or_records_get_more(monkeypatch):
    """A launch whose walkers run out of contig pool or contig records is restarted with twice as
    much of what ran out (the walkers say which: WSTAT_OVF_POOL / WSTAT_OVF_RECS) -- not with a
    larger vertex table, which is the remedy for every other overflow."""
    g = GoldenCase("k64")
    kw = g.kwargs()
    for env in ({"ABG_POOL_CAP": "3000"}, {"ABG_REC_CAP": "16"}):
        for key in ("ABG_POOL_CAP", "ABG_REC_CAP"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        hc = HostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=512)
        hc.load(g.buf, g.off)
        results, contigs = hc.assemble(g.buf, g.off)
        assert hc.stats()["overflows"] >= 2, hc.stats()
        assert api.format_fasta(contigs, g.ids) == g.fasta
        assert api.format_read_log(results, g.ids) == g.readlog
        assert api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace


def test_tiled_insert_matches_oracle_and_survives_bin_overflow(monkeypatch):
    """PASS 1 through LDS-sized tiles (TileEnv, abg_engine.h): k-mers that share no counter with
    another k-mer of the batch are settled tile by tile, the rest by the reservation rounds.  Same
    counter array as the oracle's sequential incrementMin -- also when a filter saturates, when the
    bins are too small (ABG_TILE_CAP: the batch then takes the rounds as a whole) and with the tiles
    switched off (ABG_TILED=0)."""
    k = 40
    m1, m2 = synth.make_read_set(30000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    for counters, env, want in ((1 << 21, {}, "tiled"), (1 << 21, {"ABG_TILE_CAP": "600"}, "overflow"),
                                (1 << 21, {"ABG_TILED": "0"}, "rounds"), (1 << 19, {}, "tiled")):
        for key in ("ABG_TILE_CAP", "ABG_TILED"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        o = ob.Oracle(k, counters=counters)
        hc = HostCheck(k, counters, insert_batch=30000, claim_log2=16)
        o.load(buf, off)
        hc.load(buf, off)
        st = hc.stats()
        if want == "tiled":
            assert st["tiled_ops"] > 0 and st["tile_overflows"] == 0 and 0 < st["tiled_pending"] < st["tiled_ops"]
        elif want == "overflow":
            assert st["tile_overflows"] > 0
        else:
            assert st["tiled_ops"] == 0
        assert np.array_equal(o.counters(), hc.counters()), (counters, env)
    # a repeated sequence drives counters to 255 within one batch: n ops of one k-mer in one go
    rep = np.tile(np.frombuffer(b"ACGTTGCATGCCGATAGCTAGGATCCATGCAAGCTTGGCATTCGGATACCGGTAAGCTAGCTAACGGT", dtype=np.uint8), (400, 1))
    buf, off = api.matrix_to_seqs(rep)
    o = ob.Oracle(k, counters=1 << 18)
    hc = HostCheck(k, 1 << 18, insert_batch=30000, claim_log2=16)
    o.load(buf, off)
    hc.load(buf, off)
    assert hc.counters().max() == 255 and np.array_equal(o.counters(), hc.counters())
    # homopolymer runs: one k-mer hundreds of times in a batch.  n ops of a k-mer nobody shares a counter with are n ops however
    # many: its leader raises the counters by n (until round 6 a counter with 254 pairs or more was left to the rounds, one round per op)
    buf, off = api.concat_seqs([b"A" * 150] * 15 + [bytes(r) for r in rep[:50]] + [b"AC" * 75] * 20)  # (1,665 ops of one k-mer: a bin holds them)
    o = ob.Oracle(k, counters=1 << 20)
    hc = HostCheck(k, 1 << 20, insert_batch=30000, claim_log2=16)
    o.load(buf, off)
    hc.load(buf, off)
    st = hc.stats()
    assert st["tiled_ops"] > 0 and st["tiled_pending"] == 0 and st["tile_overflows"] == 0 and st["insert_rounds"] == 0
    assert np.array_equal(o.counters(), hc.counters())


def test_a_kmer_that_recurs_thousands_of_times_in_a_batch_does_not_serialise_it(monkeypatch):
    """A homopolymer run, a two-base and a five-base satellite at several hundred-fold coverage among ordinary reads: their k-mers'
    pairs run the bins of their counters over (no ABG_TILE_CAP here).  The batch is then judged and applied through a sort of its
    pairs (Engine::sorted_judge) -- same counters as the sequential filter, and about as many reservation rounds as the same reads
    take without the repeats; with that switched off the whole batch takes the rounds, one round per copy (the old behaviour,
    still exact)."""
    k = 40
    m1, m2 = synth.make_read_set(30000, 30.0)
    plain = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    rng = np.random.default_rng(4)
    hot = [b"A" * 150] * 400 + [b"AC" * 75] * 300 + [b"CA" * 75] * 100 + [(b"GATTA" * 30)[i % 5:][:140] for i in range(300)]
    hot += [b"T" * 150] * 100  # (the homopolymer's reverse complement: the same canonical k-mer)
    mixed = plain + hot
    order = rng.permutation(len(mixed))
    reads = [mixed[i] for i in order]
    buf, off = api.concat_seqs(reads)
    pbuf, poff = api.concat_seqs(plain)
    counters = 1 << 21
    base = HostCheck(k, counters, insert_batch=1 << 20, claim_log2=16)
    base.load(pbuf, poff)
    base_rounds = base.stats()["insert_rounds"]
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    hc = HostCheck(k, counters, insert_batch=1 << 20, claim_log2=16)
    hc.load(buf, off)
    st = hc.stats()
    assert st["tile_overflows"] > 0 and st["tiled_ops"] > 0
    assert np.array_equal(o.counters(), hc.counters())
    assert st["insert_rounds"] <= 3 * max(base_rounds, 8), (st["insert_rounds"], base_rounds)
    assert o.counters().max() == 255
    # the fall-back of old: exact, and a chain of rounds as long as the hottest k-mer's copies
    monkeypatch.setenv("ABG_SORTED_OVERFLOW", "0")
    old = HostCheck(k, counters, insert_batch=1 << 20, claim_log2=16)
    old.load(buf, off)
    assert np.array_equal(o.counters(), old.counters())
    assert old.stats()["insert_rounds"] > 20 * st["insert_rounds"]


def test_kmers_writing_shared_counters_are_settled_together_or_go_to_the_rounds_together(monkeypatch):
    """op_verdict's round-5 rule (abg_engine.h): a k-mer that may raise a counter it shares is settled by the tiles when every
    k-mer on that counter is (FCoSettle's fixed point, FCoFinal), the shared counter ending at the largest target -- the
    counter array stays the oracle's sequential incrementMin (CountingBloomFilter.hpp:135-162) at three occupancies, with
    the rule off, with too few passes for the fixed point (every candidate then takes the rounds), and with a table of
    marked counters so small that nearly every candidate meets a marked bit; and it sends an order of magnitude fewer
    ops to the reservation rounds."""
    k = 40
    m1, m2 = synth.make_read_set(30000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    keys = ("ABG_COSETTLE", "ABG_COSETTLE_PASSES", "ABG_COSETTLE_LOG2")
    for counters in (1 << 21, 1 << 19, 1 << 18):
        o = ob.Oracle(k, counters=counters)
        o.load(buf, off)
        pending = {}
        for name, env in (("off", {"ABG_COSETTLE": "0"}), ("on", {}), ("one_pass", {"ABG_COSETTLE_PASSES": "1"}),
                          ("two_passes", {"ABG_COSETTLE_PASSES": "2"}), ("tiny_table", {"ABG_COSETTLE_LOG2": "10"})):
            for key in keys:
                monkeypatch.delenv(key, raising=False)
            for key, val in env.items():
                monkeypatch.setenv(key, val)
            hc = HostCheck(k, counters, insert_batch=30000, claim_log2=16)
            hc.load(buf, off)
            st = hc.stats()
            assert st["tiled_ops"] > 0 and st["tile_overflows"] == 0, (counters, name, st)
            assert np.array_equal(o.counters(), hc.counters()), (counters, name)
            pending[name] = st["tiled_pending"]
        assert pending["on"] * 10 < pending["off"], pending
        assert pending["on"] <= pending["two_passes"] <= pending["one_pass"] <= pending["off"], pending
        assert pending["on"] <= pending["tiny_table"] <= pending["off"], pending
    # counters driven into saturation while k-mers share them: 40 copies of a small read set, a dense filter
    m1, m2 = synth.make_read_set(3000, 30.0)
    reads = synth.codes_to_ascii(np.concatenate([m1, m2]))
    buf, off = api.matrix_to_seqs(np.tile(reads, (12, 1)))
    o = ob.Oracle(k, counters=1 << 16)
    hc = HostCheck(k, 1 << 16, insert_batch=30000, claim_log2=16)
    o.load(buf, off)
    hc.load(buf, off)
    assert hc.counters().max() == 255 and np.array_equal(o.counters(), hc.counters())


@pytest.mark.parametrize("nh", [1, 2, 3, 5, 8, 9])
def test_the_settling_rule_with_few_and_many_hash_functions(nh):
    """op_verdict's rules read an op's flags per hash function (a byte: up to 8) and learn the k-mer's op count from its first two
    counters: one hash function (no second counter to learn from), two, an odd number, the most a flag byte holds, and one more
    (every flagged k-mer then takes the rounds) -- the counter array stays the oracle's on a filter dense enough for chains."""
    k = 33
    m1, m2 = synth.make_read_set(20000, 30.0, genome_seed=nh, read_seed=nh + 50)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    counters = 1 << 19
    o = ob.Oracle(k, counters=counters, num_hashes=nh)
    hc = HostCheck(k, counters, num_hashes=nh, insert_batch=30000, claim_log2=16)
    o.load(buf, off)
    hc.load(buf, off)
    st = hc.stats()
    assert st["tiled_ops"] > 0, st
    assert np.array_equal(o.counters(), hc.counters())


def test_kmer_helpers_and_prefix_xor_hashes_agree_with_the_per_base_forms():
    """hc_selftest_kmer: window_kmer vs batch_kmer, kmer_revcomp_fast vs kmer_revcomp, kmer_hashes vs vtx_rehash,
    and the prefix-XOR hashes of stretches of consecutive k-mers (stretch_hashes_serial: the arithmetic of the
    device's stretch_hashes_wave, which walk_bulk / chain_bulk / FPresearchScan use) vs kmer_hashes."""
    l = C.CDLL(build.build_hostcheck())
    l.hc_selftest_kmer.restype = C.c_uint64
    l.hc_selftest_kmer.argtypes = [C.c_uint, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(5)
    L = 300
    pad = np.zeros(((L + 15) // 16) * 16, dtype=np.uint64)
    pad[:L] = rng.integers(0, 4, size=L)
    words = np.concatenate([(pad.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32), np.zeros(4, dtype=np.uint32)])
    for k in (21, 31, 32, 33, 34, 62, 63, 64, 65, 66, 67, 93, 96, 99, 100, 127, 128, 150, 192):  # (31, 33, 34, 62, 66, 93, 99 ...: a rotation amount of SeedTabs is zero)
        assert l.hc_selftest_kmer(k, words.ctypes.data, L) == 0, k


@pytest.mark.parametrize("k,mask,ahead", [(33, None, True), (40, "k40", True), (48, "K16", True), (33, None, False)])
def test_reads_kept_on_the_device_between_the_passes_assemble_like_the_read_stream(k, mask, ahead, monkeypatch):
    """abg_keep_reads / abg_load_seqs_v / abg_assemble_kept against the oracle fed the same stream:
    reads with N, short reads, lower case, a read longer than a PASS-1 piece, empty chunks, several
    load calls, the packing split over threads; a call's arrays uploaded ahead by the calling thread (the default) or by the
    library's thread in front of the chunk's kernels (ABG_NO_UPLOAD_AHEAD)."""
    monkeypatch.setenv("ABG_HOST_SPLIT", "3")
    if not ahead:
        monkeypatch.setenv("ABG_NO_UPLOAD_AHEAD", "1")
    m1, m2 = synth.make_read_set(12000, 30.0, err=0.004, genome_seed=k, read_seed=k + 5)
    reads = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    rng = np.random.default_rng(k)
    genome = synth.codes_to_ascii(synth.make_genome(12000, seed=k)[0][None, :])[0].tobytes()
    for i in rng.choice(len(reads), 40, replace=False):
        r = bytearray(reads[i])
        r[int(rng.integers(len(r)))] = ord("N")
        reads[i] = bytes(r)
    for i in rng.choice(len(reads), 30, replace=False):
        reads[i] = reads[i].lower()
    reads[17] = reads[17][:k - 1]
    reads[400] = b""
    reads[800] = genome[1000:6000]          # longer than a piece when insert_batch is small
    reads[801] = genome[2000:3000] + b"N" + genome[3001:7000]
    seed = {"k40": None, "K16": api.spaced_seed_kmer_pair(48, 16), None: None}[mask]
    cuts = [0, 0, 700, 701, 1500, 1900, 1900, len(reads)]
    chunks = [api.concat_seqs(reads[a:b]) for a, b in zip(cuts, cuts[1:])]
    buf, off = api.concat_seqs(reads)
    o = ob.Oracle(k, counters=1 << 21, mask=seed.encode()) if seed else ob.Oracle(k, counters=1 << 21)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    hc = HostCheck(k, 1 << 21, insert_batch=3000, claim_log2=14, p2_first=200, mask=seed)
    assert hc.keep_reads(True, sum(len(r) for r in reads)) == 0
    hc.load_chunks(chunks[:3])
    hc.load_chunks(chunks[3:4])
    hc.load_chunks(chunks[4:])
    assert np.array_equal(o.counters(), hc.counters())
    rc, rh, ch = hc.assemble_kept(len(reads))
    assert rc == 0
    assert np.array_equal(ro, rh)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert o.assembly_counters() == hc.assembly_counters()
    assert {1, 2} <= set(np.unique(rh).tolist())  # SHORTER_THAN_K and NON_ACGT verdicts came from loading time
    # nothing is kept any more
    rc, _, _ = hc.assemble_kept(0)
    assert rc != 0


def test_kept_reads_dropped_for_lack_of_room_leave_pass1_intact(monkeypatch):
    """The store cannot grow (ABG_KEEP_FAIL): loading goes on without it, abg_assemble_kept says ABG_EAGAIN,
    and the caller assembles from its own buffers as if nothing had been kept."""
    k = 35
    m1, m2 = synth.make_read_set(9000, 25.0, err=0.004, genome_seed=k, read_seed=k + 5)
    reads = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    chunks = [api.concat_seqs(reads[:500]), api.concat_seqs(reads[500:])]
    buf, off = api.concat_seqs(reads)
    o = ob.Oracle(k, counters=1 << 20)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    monkeypatch.setenv("ABG_KEEP_FAIL", "1")
    hc = HostCheck(k, 1 << 20, insert_batch=20000, claim_log2=14, p2_first=200)
    assert hc.keep_reads(True, 0) == 0
    hc.load_chunks(chunks[:1])   # fits the store's first allocation
    hc.load_chunks(chunks[1:])   # would have to grow it: dropped
    assert np.array_equal(o.counters(), hc.counters())
    rc, _, _ = hc.assemble_kept(len(reads))
    assert rc == _lib.ABG_EAGAIN
    rh, ch = hc.assemble_chunks(chunks)
    assert np.array_equal(ro, rh) and [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch]
    assert o.assembly_counters() == hc.assembly_counters()


def _thicket_reads(seed, unit_len, copies, flank, coverage, err, read_len=100):
    """A genome whose middle is `copies` exact copies of one unit (spread out between unique stretches) read at
    `coverage`: the collapsed repeat carries copies x coverage, so its sequencing errors recur and every vertex of
    it has error tips and bubbles beside it -- the shape the walkers' search shortcuts are for."""
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, size=unit_len, dtype=np.uint8)
    parts = []
    for _ in range(copies):
        parts.append(rng.integers(0, 4, size=flank, dtype=np.uint8))
        parts.append(unit)
    parts.append(rng.integers(0, 4, size=flank, dtype=np.uint8))
    g = np.concatenate(parts)
    n = int(len(g) * coverage / read_len)
    start = rng.integers(0, len(g) - read_len, size=n)
    reads = g[start[:, None] + np.arange(read_len)[None, :]]
    rc = rng.random(n) < 0.5
    reads[rc] = 3 - reads[rc][:, ::-1]
    e = rng.random(reads.shape) < err
    reads[e] = (reads[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
    return reads


@pytest.mark.parametrize("k,trim,nh,kc", [(32, None, 4, 2), (41, 30, 3, 2), (64, None, 4, 2)])
def test_search_shortcuts_in_a_thicket_match_the_oracle(k, trim, nh, kc, monkeypatch):
    """successor()'s shortcuts (chain_true_branches / chain_bulk): a branch is TRUE as soon as ANY solid walk of `trim`
    edges leaves it (ExtendPath.h:174-244 is an OR over paths: read-guided, then taking the first or the last neighbour at
    every fork), FALSE with its exact depth when it is a plain dead-end tip.  A collapsed 12-copy repeat at 40x with 1.5 %
    errors is nothing but forks and tips; with the device's fast memory (HC_FAST_BYTES, so that the chain searches run),
    with and without the guide, the contigs, the read log and the visited filter are the oracle's."""
    reads = _thicket_reads(5 + k, 700, 12, 300, 40.0, 0.015, read_len=110 if k > 41 else 100)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(reads))
    counters = 1 << 22
    o = ob.Oracle(k, counters=counters, num_hashes=nh, min_cov=kc, trim=trim)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    monkeypatch.setenv("HC_FAST_BYTES", "20480")
    for stride in ("2", "0"):
        monkeypatch.setenv("ABG_GUIDE_STRIDE", stride)
        hc = HostCheck(k, counters, nh, kc, trim, insert_batch=50000, claim_log2=16, p2_first=256)
        hc.load(buf, off)
        assert np.array_equal(o.counters(), hc.counters())
        rh, ch = hc.assemble(buf, off)
        assert np.array_equal(ro, rh), stride
        assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch], stride
        assert np.array_equal(o.visited(), hc.visited())
        if stride != "0":
            assert hc.stats()["chain_steps"] > 0
