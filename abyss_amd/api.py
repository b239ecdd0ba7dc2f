"""Python mirror of the abyss-bloom-dbg unitig stage over the C ABI.

Names follow the reference: a counting Bloom filter of solid k-mers is loaded from reads
(PASS 1, BloomDBG/BloomIO.h), then reads are extended into unitigs (PASS 2,
BloomDBG/bloom-dbg.h ``assemble``), and contigs are printed as
``>ID LEN COV read:READID`` FASTA records (bloom-dbg.h:455-487).  All compute happens in
libabyss_amd.so on the GPU; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

READ_RESULT_NAMES = ["NA", "SHORTER_THAN_K", "NON_ACGT", "BLUNT_END", "NOT_SOLID", "ALL_KMERS_VISITED",
                     "ALL_BRANCH_KMERS_VISITED", "GENERATED_CONTIGS"]  # bloom-dbg.h:268-293
EXT_CODE_NAMES = ["AMBI_IN", "AMBI_OUT", "DEAD_END", "CYCLE", "LENGTH_LIMIT"]  # ExtendPath.h:62-79
NO_CONTIG = 2 ** 64 - 1


class AbyssAmdError(RuntimeError):
    pass


@dataclass
class ContigRecord:  # ContigRecord, bloom-dbg.h:186-254
    contig_id: int
    read_index: int
    seq: bytes
    coverage: int
    redundant: bool
    left_ext: int
    right_ext: int
    left_code: int
    right_code: int
    seed_pos: int


def concat_seqs(seqs: Sequence[bytes]) -> Tuple[bytes, np.ndarray]:
    """Concatenate sequences into (buffer, offsets[n+1]) as the C ABI takes them."""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return b"".join(seqs), off


def matrix_to_seqs(ascii_matrix: np.ndarray) -> Tuple[bytes, np.ndarray]:
    """An [n, L] uint8 ASCII matrix as (buffer, offsets)."""
    n, L = ascii_matrix.shape
    return np.ascontiguousarray(ascii_matrix).tobytes(), np.arange(n + 1, dtype=np.uint64) * np.uint64(L)


def spaced_seed_kmer_pair(k: int, K: int) -> str:
    """SpacedSeed::kmerPair (BloomDBG/SpacedSeed.h:18-26): `-K`."""
    assert K <= k // 2
    return "1" * K + "0" * (k - 2 * K) + "1" * K


def spaced_seed_qr(length: int) -> str:
    """SpacedSeed::qrSeed (SpacedSeed.h:40-52): '0' at the quadratic residues mod `length`."""
    residues = {j * j % length for j in range(1, length)}
    return "".join("0" if i in residues else "1" for i in range(length))


def spaced_seed_qr_pair(k: int, length: int) -> str:
    """SpacedSeed::qrSeedPair (SpacedSeed.h:64-73): `--qr-seed`."""
    assert length <= k // 2
    q = spaced_seed_qr(length)
    return q + "0" * (k - 2 * length) + q[::-1]


class BloomDBG:
    """One assembly: solid counting filter + visited filter + counters on one GPU."""

    def __init__(self, k: int, bloom_bytes: int = 0, counters: int = 0, num_hashes: int = 4, min_cov: int = 2,
                 trim: Optional[int] = None, device: int = 0, verbose: int = 0, spaced_seed: Optional[str] = None,
                 **tuning):
        self._lib = _lib.load()
        p = _lib.Params()
        self._lib.abg_params_init(C.byref(p))
        p.k, p.num_hashes, p.min_cov = k, num_hashes, min_cov
        p.trim = 0xFFFFFFFF if trim is None else trim
        p.bloom_bytes, p.counters, p.device, p.verbose = bloom_bytes, counters, device, verbose
        self._seed = spaced_seed.encode() if spaced_seed else None  # must outlive abg_create
        p.spaced_seed = self._seed
        for key, val in tuning.items():
            setattr(p, key, val)
        self._ctx = C.c_void_p()
        rc = self._lib.abg_create(C.byref(p), C.byref(self._ctx))
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_last_error(None)
            self._ctx = None
            raise AbyssAmdError("abg_create failed (%d): %s" % (rc, msg.decode() if msg else ""))
        self.k = k
        self.num_hashes = num_hashes

    def close(self):
        if getattr(self, "_ctx", None):
            self.free_device()
            self._lib.abg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_last_error(self._ctx)
            raise AbyssAmdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def reset(self) -> None:
        """Empty filters and zero counters, keeping the device memory (abg_reset)."""
        self._check(self._lib.abg_reset(self._ctx), "abg_reset")

    # ---- filter
    @property
    def size(self) -> int:
        n = C.c_uint64()
        self._check(self._lib.abg_filter_size(self._ctx, C.byref(n)), "abg_filter_size")
        return n.value

    def load(self, buf: bytes, offsets: np.ndarray) -> None:
        """PASS 1: loadSeq over every sequence in order (BloomIO.h:32-41)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._lib.abg_load_seqs(self._ctx, buf, offsets.ctypes.data, len(offsets) - 1), "abg_load_seqs")

    def load_chunks(self, chunks) -> None:
        """PASS 1 over several (buf, offsets) chunks in one call (abg_load_seqs_v)."""
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for _, o in chunks]
        nc = len(chunks)
        seqs_v = (C.c_char_p * nc)(*[C.c_char_p(b) for b, _ in chunks])
        off_v = (C.c_void_p * nc)(*[o.ctypes.data for o in offs])
        n_v = (C.c_uint64 * nc)(*[len(o) - 1 for o in offs])
        self._check(self._lib.abg_load_seqs_v(self._ctx, nc, C.cast(seqs_v, C.c_void_p), C.cast(off_v, C.c_void_p),
                                              C.cast(n_v, C.c_void_p)), "abg_load_seqs_v")

    def keep_reads(self, on: bool = True, expected_bases: int = 0) -> None:
        """The reads of the load calls that follow stay on the device, packed, for assemble_kept (abg_keep_reads)."""
        self._check(self._lib.abg_keep_reads(self._ctx, int(on), expected_bases), "abg_keep_reads")

    def assemble_kept(self, n: int) -> Tuple[np.ndarray, List[ContigRecord]]:
        """PASS 2 over the n reads loaded since keep_reads(), as one read stream (abg_assemble_kept)."""
        results = np.zeros(max(n, 1), dtype=np.uint8)
        contigs: List[ContigRecord] = []
        cb = self._collector(contigs)
        self._check(self._lib.abg_assemble_kept(self._ctx, results.ctypes.data, cb, None), "abg_assemble_kept")
        return results[:n], contigs

    def load_packed(self, words_ptr: int, woff_ptr: int, len_ptr: int, n: int) -> None:
        self._check(self._lib.abg_load_packed(self._ctx, words_ptr, woff_ptr, len_ptr, n), "abg_load_packed")

    # ---- partitioned multi-GPU run (abyss_amd.dist)
    def attach_comm(self, comm) -> None:
        """Range-partition the counting filter over the ranks of `comm` (an abyss_amd.dist communicator).
        `comm` must stay alive as long as this object."""
        self._comm = comm
        self._check(self._lib.abg_attach_comm(self._ctx, C.byref(comm.struct)), "abg_attach_comm")

    def share_reads(self, words_ptr: int, woff_ptr: int, len_ptr: int, n: int) -> Tuple[int, int, int, int]:
        """All-gather of the ranks' packed reads: (words, woff, len) device pointers and the total count."""
        gw, go, gl, nt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self._lib.abg_share_reads(self._ctx, words_ptr, woff_ptr, len_ptr, n, C.byref(gw), C.byref(go),
                                              C.byref(gl), C.byref(nt)), "abg_share_reads")
        return gw.value, go.value, gl.value, nt.value

    def to_device(self, arr: np.ndarray) -> int:
        """Copy a host array into device memory of this context's GPU (abg_dev_alloc + abg_dev_copy);
        returns the device pointer, released by free_device() or close()."""
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._check(self._lib.abg_dev_alloc(self._ctx, max(arr.nbytes, 1), C.byref(p)), "abg_dev_alloc")
        self._check(self._lib.abg_dev_copy(self._ctx, p, arr.ctypes.data, arr.nbytes, 0), "abg_dev_copy")
        self._dev = getattr(self, "_dev", [])
        self._dev.append(p.value)
        return p.value

    def free_device(self) -> None:
        for p in getattr(self, "_dev", []):
            self._lib.abg_dev_free(self._ctx, p)
        self._dev = []

    def counting_stats(self) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.abg_counting_stats(self._ctx, C.byref(a), C.byref(b)), "abg_counting_stats")
        return a.value, b.value

    def counters(self) -> np.ndarray:
        out = np.empty(self.size, dtype=np.uint8)
        self._check(self._lib.abg_counters_export(self._ctx, out.ctypes.data), "abg_counters_export")
        return out

    def set_counters_array(self, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert arr.size == self.size
        self._check(self._lib.abg_counters_import(self._ctx, arr.ctypes.data), "abg_counters_import")

    def cascade_level(self, level: int) -> np.ndarray:
        out = np.empty(self.size // 8, dtype=np.uint8)
        self._check(self._lib.abg_cascade_export(self._ctx, level, out.ctypes.data), "abg_cascade_export")
        return out

    def visited(self) -> np.ndarray:
        out = np.empty(self.size // 8, dtype=np.uint8)
        self._check(self._lib.abg_visited_export(self._ctx, out.ctypes.data), "abg_visited_export")
        return out

    def set_visited_array(self, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert arr.size == self.size // 8
        self._check(self._lib.abg_visited_import(self._ctx, arr.ctypes.data), "abg_visited_import")

    # ---- assembly
    def _collector(self, out: List[ContigRecord]):
        def cb(_user, c):
            c = c.contents
            out.append(ContigRecord(c.contig_id, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                    c.right_ext, c.left_code, c.right_code, c.seed_pos))
        return _lib.CONTIG_CB(cb)

    def assemble(self, buf: bytes, offsets: np.ndarray) -> Tuple[np.ndarray, List[ContigRecord]]:
        """PASS 2: processRead over every read in order (bloom-dbg.h:781-882)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        results = np.zeros(n, dtype=np.uint8)
        contigs: List[ContigRecord] = []
        cb = self._collector(contigs)
        self._check(self._lib.abg_assemble_seqs(self._ctx, buf, offsets.ctypes.data, n, results.ctypes.data, cb, None),
                    "abg_assemble_seqs")
        return results, contigs

    def assemble_chunks(self, chunks) -> Tuple[np.ndarray, List[ContigRecord]]:
        """PASS 2 over a read set held in several (buf, offsets) chunks, as ONE pass (abg_assemble_seqs_v);
        read indices count through the chunks in order."""
        bufs = [C.c_char_p(b) for b, _ in chunks]
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for _, o in chunks]
        nc = len(chunks)
        seqs_v = (C.c_char_p * nc)(*bufs)
        off_v = (C.c_void_p * nc)(*[o.ctypes.data for o in offs])
        n_v = (C.c_uint64 * nc)(*[len(o) - 1 for o in offs])
        results = np.zeros(sum(len(o) - 1 for o in offs), dtype=np.uint8)
        contigs: List[ContigRecord] = []
        cb = self._collector(contigs)
        self._check(self._lib.abg_assemble_seqs_v(self._ctx, nc, C.cast(seqs_v, C.c_void_p), C.cast(off_v, C.c_void_p),
                                                  C.cast(n_v, C.c_void_p), results.ctypes.data, cb, None), "abg_assemble_seqs_v")
        return results, contigs

    def assemble_packed(self, words_ptr: int, woff_ptr: int, len_ptr: int, n: int, want_results: bool = True,
                        want_contigs: bool = True) -> Tuple[Optional[np.ndarray], List[ContigRecord]]:
        results = np.zeros(n, dtype=np.uint8) if want_results else None
        contigs: List[ContigRecord] = []
        cb = self._collector(contigs) if want_contigs else _lib.CONTIG_CB()  # NULL: counters only
        self._check(self._lib.abg_assemble_packed(self._ctx, words_ptr, woff_ptr, len_ptr, n,
                                                  results.ctypes.data if want_results else None, cb, None),
                    "abg_assemble_packed")
        return results, contigs

    def output_graph(self, buf: bytes, offsets: np.ndarray, frame: bool = True) -> Tuple[bytes, int, int]:
        """-g: outputGraph (bloom-dbg.h:1171-1242) over these sequences: (GraphViz text, nodes, edges).
        The visited-vertex set carries over between calls; frame=False leaves out "digraph g {" / "}"."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        parts = [b"digraph g {\n"] if frame else []
        cb = _lib.TEXT_CB(lambda _u, p, n: parts.append(C.string_at(p, n)))
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.abg_output_graph_seqs(self._ctx, buf, offsets.ctypes.data, len(offsets) - 1, cb, None,
                                                    C.byref(a), C.byref(b)), "abg_output_graph_seqs")
        if frame:
            parts.append(b"}\n")
        return b"".join(parts), a.value, b.value

    def assembly_counters(self) -> dict:
        c = _lib.Counters()
        self._check(self._lib.abg_get_counters(self._ctx, C.byref(c)), "abg_get_counters")
        return {n: getattr(c, n) for n, _ in c._fields_}

    def stats(self) -> dict:
        s = _lib.Stats()
        self._check(self._lib.abg_get_stats(self._ctx, C.byref(s)), "abg_get_stats")
        return {n: getattr(s, n) for n, _ in s._fields_}

    # ---- probes / profiling
    def hash_seq(self, seq: bytes) -> Tuple[np.ndarray, np.ndarray]:
        cap = max(len(seq), 1)
        pos = np.zeros(cap, dtype=np.uint32)
        hashes = np.zeros((cap, self.num_hashes), dtype=np.uint64)
        n = C.c_uint64()
        self._check(self._lib.abg_hash_seq(self._ctx, seq, len(seq), pos.ctypes.data, hashes.ctypes.data, cap,
                                           C.byref(n)), "abg_hash_seq")
        return pos[:n.value], hashes[:n.value]

    def contains_seq(self, seq: bytes) -> Tuple[np.ndarray, np.ndarray]:
        """(positions, 0/1) of the valid k-mers of `seq` in the solid filter (writeCovTrack, bloom-dbg.h:1282-1334)."""
        cap = max(len(seq), 1)
        pos = np.zeros(cap, dtype=np.uint32)
        val = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint64()
        self._check(self._lib.abg_contains_seq(self._ctx, seq, len(seq), pos.ctypes.data, val.ctypes.data, cap,
                                               C.byref(n)), "abg_contains_seq")
        return pos[:n.value], val[:n.value]

    def profile_enable(self, on: bool = True) -> None:
        self._check(self._lib.abg_profile_enable(self._ctx, int(on)), "abg_profile_enable")

    def profile_reset(self) -> None:
        self._check(self._lib.abg_profile_reset(self._ctx), "abg_profile_reset")

    def profile_get(self, name: str) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.abg_profile_get(self._ctx, name.encode(), C.byref(ms), C.byref(n)), "abg_profile_get")
        return ms.value, n.value


def pack_ends(seqs: Sequence[bytes], overlap: int) -> Tuple[np.ndarray, np.ndarray]:
    """First and last `overlap` bases of every sequence as the 2-bit keys abg_overlap_join takes
    (Kmer(seq.substr(...)), AdjList.cpp:210-211): [n, W] uint64, base j at bits 2(j%32) of word j/32."""
    n, W = len(seqs), (overlap + 31) // 32
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    out = []
    for take in (lambda s: s[:overlap], lambda s: s[len(s) - overlap:]):
        ends = np.frombuffer(b"".join(take(s) for s in seqs), dtype=np.uint8).reshape(n, overlap) if n else np.zeros((0, overlap), np.uint8)
        c = code[ends]
        if (c == 255).any():
            raise ValueError("unexpected character in a contig end")
        pad = np.zeros((n, W * 32), dtype=np.uint64)
        pad[:, :overlap] = c
        out.append(np.ascontiguousarray((pad.reshape(n, W, 32) << (2 * np.arange(32, dtype=np.uint64))).sum(axis=2, dtype=np.uint64)))
    return out[0], out[1]


class OverlapJoin:
    """AdjList's join of contig ends overlapping by exactly k-1 bases, on one GPU
    (buildOverlapGraph, AdjList/AdjList.cpp:233-263; include/abyss_amd.h abg_overlap_*)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.abg_overlap_create(device, C.byref(self._h))
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_overlap_last_error(None)
            self._h = None
            raise AbyssAmdError("abg_overlap_create failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.abg_overlap_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_overlap_last_error(self._h)
            raise AbyssAmdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def join(self, overlap: int, head: np.ndarray, tail: np.ndarray, strand_specific: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """(offsets [2n+1], targets): out-edges of vertex s (2i = i+, 2i+1 = i-) are targets[offsets[s]:offsets[s+1]]."""
        head = np.ascontiguousarray(head, dtype=np.uint64)
        tail = np.ascontiguousarray(tail, dtype=np.uint64)
        n = head.shape[0] if head.ndim == 2 else 0
        ne = C.c_uint64()
        self._check(self._lib.abg_overlap_join(self._h, overlap, n, head.ctypes.data, tail.ctypes.data, int(strand_specific), C.byref(ne)),
                    "abg_overlap_join")
        off = np.zeros(2 * n + 1, dtype=np.uint64)
        tgt = np.zeros(ne.value, dtype=np.uint32)
        self._check(self._lib.abg_overlap_edges(self._h, off.ctypes.data, tgt.ctypes.data if ne.value else None), "abg_overlap_edges")
        return off, tgt

    def profile(self, on: bool = True) -> None:
        self._lib.abg_overlap_profile(self._h, int(on))

    def profile_get(self, name: str) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        self._lib.abg_overlap_profile_get(self._h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value


def counting_bloom_file(counters: np.ndarray, k: int, num_hashes: int) -> bytes:
    """`operator<<` of CountingBloomFilter<uint8_t> (CountingBloomFilter.hpp:344-379): the TOML-ish
    header in the key order cpptoml emits, then the raw counters."""
    n = int(counters.size)
    head = ("[BTLCountingBloomFilter_v1]\n\tBloomFilterSize = %d\n\tHashNum = %d\n\tKmerSize = %d\n"
            "\tBloomFilterSizeInBytes = %d\n\tBitsPerCounter = 8\n[HeaderEnd]\n" % (n, num_hashes, k, n))
    return head.encode() + np.ascontiguousarray(counters, dtype=np.uint8).tobytes()


def bit_bloom_file(bits: np.ndarray, k: int, num_hashes: int) -> bytes:
    """`operator<<` of BloomFilter (BloomFilter.hpp:261-294) for a filter of bits.size * 8 bits."""
    nbytes = int(bits.size)
    head = ("[BTLBloomFilter_v1]\n\tnEntry = 0\n\tdFPR = 0.0000000000000000\n\tEntry = 0\n"
            "\tBloomFilterSizeInBytes = %d\n\tBloomFilterSize = %d\n\tHashNum = %d\n\tKmerSize = %d\n[HeaderEnd]\n"
            % (nbytes, nbytes * 8, num_hashes, k))
    return head.encode() + np.ascontiguousarray(bits, dtype=np.uint8).tobytes()


def format_fasta(contigs: Iterable[ContigRecord], read_ids: Sequence[bytes]) -> bytes:
    """printContig (bloom-dbg.h:455-487): ``>ID LEN COV read:READID`` + sequence, non-redundant only."""
    out = []
    for c in contigs:
        if c.redundant:
            continue
        out.append(b">%d %d %d read:%s\n%s\n" % (c.contig_id, len(c.seq), c.coverage, read_ids[c.read_index], c.seq))
    return b"".join(out)


def format_trace(contigs: Iterable[ContigRecord], read_ids: Sequence[bytes], reads: Sequence[bytes], k: int,
                 with_length: bool = True) -> bytes:
    """-T trace rows (ContigRecord operator<<, bloom-dbg.h:229-254).  The reference leaves `length`
    uninitialised for redundant contigs, so parity checks drop that column (with_length=False)."""
    rows = [b"contig_id\tlength\tredundant\tread_id\tleft_result\tleft_extension\tright_result\t"
            b"right_extension\tseed_type\tseed_length\tseed\n" if with_length else
            b"contig_id\tredundant\tread_id\tleft_result\tleft_extension\tright_result\t"
            b"right_extension\tseed_type\tseed_length\tseed\n"]
    for c in contigs:
        f = [b"NA" if c.redundant else b"%d" % c.contig_id]
        if with_length:
            f.append(b"%d" % len(c.seq))
        f += [b"%d" % int(c.redundant), read_ids[c.read_index]]
        f += [EXT_CODE_NAMES[c.left_code].encode(), b"%d" % c.left_ext] if c.left_ext > 0 else [b"NA", b"NA"]
        f += [EXT_CODE_NAMES[c.right_code].encode(), b"%d" % c.right_ext] if c.right_ext > 0 else [b"NA", b"NA"]
        seed = reads[c.read_index][c.seed_pos:c.seed_pos + k].upper()
        f += [b"READ", b"%d" % k, seed]
        rows.append(b"\t".join(f) + b"\n")
    return b"".join(rows)


def format_read_log(results: np.ndarray, read_ids: Sequence[bytes]) -> bytes:
    """--read-log rows (ReadRecord, bloom-dbg.h:300-334)."""
    rows = [b"read_id\tresult\n"]
    for rid, r in zip(read_ids, results):
        rows.append(rid + b"\t" + READ_RESULT_NAMES[int(r)].encode() + b"\n")
    return b"".join(rows)


class ReadFilter:
    """The Bloom filter of the reads' r-mers that abyss-rresolver-short keeps per r value, on one GPU
    (btllib::KmerBloomFilter as RResolver/BloomFilters.cpp:139-264 uses it; include/abyss_amd.h abg_rr_*)."""

    def __init__(self, nbytes: int, r: int, hash_num: int = 7, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.abg_rr_create(device, nbytes, hash_num, r, C.byref(self._h))
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_rr_last_error(None)
            self._h = None
            raise AbyssAmdError("abg_rr_create failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.abg_rr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.ABG_OK:
            msg = self._lib.abg_rr_last_error(self._h)
            raise AbyssAmdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    @property
    def nbytes(self) -> int:
        b = C.c_uint64()
        self._check(self._lib.abg_rr_bytes(self._h, C.byref(b)), "abg_rr_bytes")
        return b.value

    def clear(self) -> None:
        self._check(self._lib.abg_rr_clear(self._h), "abg_rr_clear")

    def insert(self, buf: bytes, off: np.ndarray, max_bases: int, lengths: Sequence[int] = ()) -> int:
        """insert(seq[:max_bases]) for every sequence of one of `lengths` (all when empty); returns how many were of a wanted length."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        n = C.c_uint64(0)
        keep = C.c_char_p(buf)
        self._check(self._lib.abg_rr_insert_seqs(self._h, C.cast(keep, C.c_void_p), off.ctypes.data, len(off) - 1, max_bases,
                                                 ln.ctypes.data if len(ln) else None, len(ln), C.byref(n)), "abg_rr_insert_seqs")
        return n.value

    def contains(self, buf: bytes, off: np.ndarray) -> np.ndarray:
        """How many r-mers of every sequence the filter holds."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        out = np.zeros(len(off) - 1, dtype=np.uint32)
        keep = C.c_char_p(buf)
        self._check(self._lib.abg_rr_contains_seqs(self._h, C.cast(keep, C.c_void_p), off.ctypes.data, len(off) - 1, out.ctypes.data if len(out) else None),
                    "abg_rr_contains_seqs")
        return out

    def popcount(self) -> int:
        c = C.c_uint64()
        self._check(self._lib.abg_rr_popcount(self._h, C.byref(c)), "abg_rr_popcount")
        return c.value

    def export(self) -> np.ndarray:
        out = np.zeros(self.nbytes, dtype=np.uint8)
        self._check(self._lib.abg_rr_export(self._h, out.ctypes.data), "abg_rr_export")
        return out

    def profile(self, on: bool = True) -> None:
        self._lib.abg_rr_profile(self._h, int(on))

    def profile_get(self, name: str) -> Tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        self._lib.abg_rr_profile_get(self._h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value
