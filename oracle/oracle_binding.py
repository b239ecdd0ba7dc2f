"""ctypes binding of oracle/liboracle.so (the CPU restatement) and helpers around
oracle/_ref (the unmodified reference built from /root/reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the abyss_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
CLI = os.path.join(HERE, "abg_oracle")
REF_BIN = os.path.join(HERE, "_ref", "abyss-bloom-dbg")
REF_TIER1 = os.path.join(HERE, "_ref", "tier1")


class OContig(C.Structure):
    _fields_ = [
        ("contig_id", C.c_uint64), ("read_index", C.c_uint64), ("seq", C.c_char_p), ("length", C.c_uint32),
        ("coverage", C.c_uint32), ("redundant", C.c_int), ("left_ext", C.c_uint32), ("right_ext", C.c_uint32),
        ("left_code", C.c_int), ("right_code", C.c_int), ("seed", C.c_char_p),
    ]


OCB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(OContig))
_lib = None


TEXT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_size_t)


def ensure_built() -> None:
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        subprocess.run(["make", "-C", HERE, "oracle"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        ensure_built()
        l = C.CDLL(LIB)
        vp = C.c_void_p
        l.orc_counters_for_budget.restype = C.c_uint64
        l.orc_counters_for_budget.argtypes = [C.c_uint64]
        l.orc_create.restype = vp
        l.orc_create.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_char_p]
        l.orc_destroy.argtypes = [vp]
        l.orc_destroy.restype = None
        l.orc_size.restype = C.c_uint64
        l.orc_size.argtypes = [vp]
        l.orc_counters.restype = C.POINTER(C.c_uint8)
        l.orc_counters.argtypes = [vp]
        l.orc_visited.restype = C.POINTER(C.c_uint8)
        l.orc_visited.argtypes = [vp]
        l.orc_popcount.restype = C.c_uint64
        l.orc_popcount.argtypes = [vp]
        l.orc_filtered_popcount.restype = C.c_uint64
        l.orc_filtered_popcount.argtypes = [vp]
        l.orc_hash_seq.restype = C.c_uint64
        l.orc_hash_seq.argtypes = [vp, C.c_char_p, C.c_size_t, vp, vp, C.c_uint64]
        l.orc_load_seqs.argtypes = [vp, C.c_char_p, vp, C.c_uint64]
        l.orc_load_seqs.restype = None
        l.orc_insert_hashes.argtypes = [vp, vp, C.c_uint64]
        l.orc_min_count.argtypes = [vp, vp, C.c_uint64, vp]
        l.orc_assemble.restype = C.c_uint64
        l.orc_assemble.argtypes = [vp, C.c_char_p, vp, C.c_uint64, vp, OCB, vp]
        l.orc_counters_get.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 5
        l.orc_look_ahead.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint]
        l.orc_successor.argtypes = [vp, C.c_char_p, C.c_int, C.c_uint, C.c_uint, C.c_char_p]
        l.orc_out_mask.argtypes = [vp, C.c_char_p]
        l.orc_in_mask.argtypes = [vp, C.c_char_p]
        l.orc_output_graph.argtypes = [vp, C.c_char_p, vp, C.c_uint64, TEXT_CB, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        l.orc_output_graph.restype = None
        l.orc_seed_kmer_pair.argtypes = [C.c_uint, C.c_uint, C.c_char_p]
        l.orc_seed_qr.argtypes = [C.c_uint, C.c_char_p]
        l.orc_seed_qr_pair.argtypes = [C.c_uint, C.c_uint, C.c_char_p]
        _lib = l
    return _lib


class Oracle:
    """CPU restatement with the same surface as abyss_amd.api.BloomDBG."""

    def __init__(self, k, bloom_bytes=0, counters=0, num_hashes=4, min_cov=2, trim=None, mask: Optional[bytes] = None):
        l = lib()
        if not counters:
            counters = l.orc_counters_for_budget(bloom_bytes)
        self.k, self.num_hashes = k, num_hashes
        self._h = l.orc_create(k, num_hashes, min_cov, k if trim is None else trim, counters, mask)
        if not self._h:
            raise ValueError("orc_create rejected the parameters")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return lib().orc_size(self._h)

    def load(self, buf: bytes, offsets: np.ndarray):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lib().orc_load_seqs(self._h, buf, offsets.ctypes.data, len(offsets) - 1)

    def counting_stats(self):
        return lib().orc_popcount(self._h), lib().orc_filtered_popcount(self._h)

    def counters(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_counters(self._h), (self.size,)).copy()

    def visited(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_visited(self._h), (self.size // 8,)).copy()

    def hash_seq(self, seq: bytes):
        cap = max(len(seq), 1)
        pos = np.zeros(cap, dtype=np.uint32)
        hashes = np.zeros((cap, self.num_hashes), dtype=np.uint64)
        n = lib().orc_hash_seq(self._h, seq, len(seq), pos.ctypes.data, hashes.ctypes.data, cap)
        return pos[:n], hashes[:n]

    def min_count(self, hashes: np.ndarray) -> np.ndarray:
        """CountingBloomFilter::minCount for rows of num_hashes hash values (CountingBloomFilter.hpp:172-182)."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.zeros(len(hashes), dtype=np.uint8)
        lib().orc_min_count(self._h, hashes.ctypes.data, len(hashes), out.ctypes.data)
        return out

    def assemble(self, buf: bytes, offsets: np.ndarray):
        from abyss_amd.api import ContigRecord  # plain dataclass, no GPU code involved
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        results = np.zeros(n, dtype=np.uint8)
        out: List = []
        seeds: List[bytes] = []

        def cb(_u, c):
            c = c.contents
            cid = c.contig_id
            # seed_pos is not reported by the oracle; recover it from the seed string
            out.append(ContigRecord(cid, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                    c.right_ext, c.left_code, c.right_code, -1))
            seeds.append(c.seed[:self.k])

        lib().orc_assemble(self._h, buf, offsets.ctypes.data, n, results.ctypes.data, OCB(cb), None)
        for rec, seed in zip(out, seeds):
            a, b = int(offsets[rec.read_index]), int(offsets[rec.read_index + 1])
            rec.seed_pos = buf[a:b].upper().find(seed)
        return results, out

    def assembly_counters(self) -> dict:
        v = [C.c_uint64() for _ in range(5)]
        lib().orc_counters_get(self._h, *[C.byref(x) for x in v])
        names = ("solid_reads", "visited_reads", "reads_processed", "bases_assembled", "next_contig_id")
        return {n: x.value for n, x in zip(names, v)}

    def output_graph(self, buf: bytes, offsets: np.ndarray):
        """-g (outputGraph, bloom-dbg.h:1171-1242): (GraphViz text incl. the digraph frame, nodes, edges)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        parts = [b"digraph g {\n"]
        cb = TEXT_CB(lambda _u, p, n: parts.append(C.string_at(p, n)))
        a, b = C.c_uint64(), C.c_uint64()
        lib().orc_output_graph(self._h, buf, offsets.ctypes.data, len(offsets) - 1, cb, None, C.byref(a), C.byref(b))
        parts.append(b"}\n")
        return b"".join(parts), a.value, b.value

    def look_ahead(self, kmer: bytes, direction: int, depth: int) -> bool:
        return bool(lib().orc_look_ahead(self._h, kmer, direction, depth))

    def successor(self, kmer: bytes, direction: int, trim: int, fp_trim: int = 5):
        out = C.create_string_buffer(self.k + 1)
        code = lib().orc_successor(self._h, kmer, direction, trim, fp_trim, out)
        return code, out.value

    def out_mask(self, kmer: bytes) -> int:
        return lib().orc_out_mask(self._h, kmer)

    def in_mask(self, kmer: bytes) -> int:
        return lib().orc_in_mask(self._h, kmer)


def have_ref() -> bool:
    return os.path.exists(REF_BIN)


def run_ref(args: Sequence[str], cwd: str, threads: int = 1, timeout: float = 3600) -> Tuple[bytes, bytes]:
    """Run the unmodified reference binary (oracle/_ref/abyss-bloom-dbg); returns (stdout, stderr)."""
    r = subprocess.run([REF_BIN, "-j%d" % threads] + list(args), cwd=cwd, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("reference failed: %s" % r.stderr.decode(errors="replace")[-2000:])
    return r.stdout, r.stderr
