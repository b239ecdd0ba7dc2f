"""Worker of tests/test_dist_partition.py: one rank of a partitioned run on the CPU.

Every rank drives a tests/hostcheck session (the product's device logic executed serially)
joined to the others by abyss_amd.dist.StagedTorchComm over gloo; rank 0 also runs the same
input through the oracle (or compares with a golden reference run) and prints one JSON line.
TEST INFRASTRUCTURE: launched with torch.distributed.run, world_size 2 or 3.
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle_binding as ob  # noqa: E402
from abyss_amd import _lib, api, dist as adist, synth  # noqa: E402
from test_hostcheck import HostCheck  # noqa: E402
from util import GoldenCase, contig_tuple, mask_of  # noqa: E402


class DistHostCheck(HostCheck):
    def attach(self):
        self.comm = adist.StagedTorchComm(*adist.host_memory_io())
        if os.environ.get("ABG_TEST_OLD_COMM_ABI"):
            # a caller compiled against the header whose abg_comm ended after all_reduce: struct_size is a zeroed field there, and
            # what lies behind the struct is not a function pointer
            self.comm.struct.struct_size = 0
            self.comm.struct.all_to_all_v = adist.A2A_FN(0)
        self.l.hc_attach_comm.argtypes = [C.c_void_p, C.c_void_p]
        assert self.l.hc_attach_comm(self.h, C.byref(self.comm.struct)) == 0
        vp = C.c_void_p
        self.l.hc_share_reads.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
        self.l.hc_load_packed.argtypes = [vp, vp, vp, vp, C.c_uint64]
        self.l.hc_assemble_packed.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, _lib.CONTIG_CB, vp]

    def share(self, words, woff, lens):
        gw, go, gl, nt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        assert self.l.hc_share_reads(self.h, words.ctypes.data, woff.ctypes.data, lens.ctypes.data, len(lens),
                                     C.byref(gw), C.byref(go), C.byref(gl), C.byref(nt)) == 0
        return gw, go, gl, nt.value

    def load_packed(self, gw, go, gl, n):
        assert self.l.hc_load_packed(self.h, gw, go, gl, n) == 0

    def assemble_packed(self, gw, go, gl, n):
        res = np.zeros(n, dtype=np.uint8)
        out = []

        def cb(_u, c):
            c = c.contents
            out.append(api.ContigRecord(c.contig_id, c.read_index, c.seq, c.coverage, bool(c.redundant), c.left_ext,
                                        c.right_ext, c.left_code, c.right_code, c.seed_pos))
        assert self.l.hc_assemble_packed(self.h, gw, go, gl, n, res.ctypes.data, _lib.CONTIG_CB(cb), None) == 0
        return res, out


def pack(codes):
    """[n, L] base codes 0..3 -> (words, woff, len) in the packed layout of include/abyss_amd.h."""
    n, L = codes.shape
    wpr = (L + 15) // 16
    pad = np.zeros((n, wpr * 16), dtype=np.uint64)
    pad[:, :L] = codes
    words = (pad.reshape(n, wpr, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=2).astype(np.uint32)
    return np.ascontiguousarray(words.reshape(-1)), np.arange(n + 1, dtype=np.uint64) * np.uint64(wpr), np.full(n, L, dtype=np.uint32)


def case_golden(name, rank, world):
    g = GoldenCase(name)
    kw = g.kwargs()
    hc = DistHostCheck(kw["k"], g.meta["counters"], kw["num_hashes"], kw["min_cov"], kw["trim"], insert_batch=50000,
                       claim_log2=16, p2_first=128, mask=mask_of(g))
    hc.attach()
    hc.load(g.buf, g.off)
    pass1 = dict(hc.comm.calls)  # what PASS 1 sent through the communicator (buffer sizes handed to the collectives)
    fp = hc.counting_stats()[1]
    results, contigs = hc.assemble(g.buf, g.off)
    c = hc.assembly_counters()
    ok = {
        "comm_pass1": pass1, "kmer_ops": int(sum(max(0, len(r) - kw["k"] + 1) for r in g.reads)),
        "filtered_popcount": fp == g.meta["filtered_popcount"],
        "fasta": api.format_fasta(contigs, g.ids) == g.fasta,
        "readlog": api.format_read_log(results, g.ids) == g.readlog,
        "trace": api.format_trace(contigs, g.ids, g.reads, g.opts["k"], with_length=False) == g.trace,
        "counters": (c["reads_processed"], c["solid_reads"], c["visited_reads"]) == (g.meta["reads"], g.meta["solid_reads"], g.meta["visited_reads"]),
    }
    return ok, hc


def case_oracle(k, G, counters, cov, err, insert_batch, rank, world, shared, p2_first=64):
    """Synthetic reads against the oracle.  shared: each rank holds a slice of the packed reads and
    the ranks all-gather them (abg_share_reads) instead of every rank passing the whole set."""
    m1, m2 = synth.make_read_set(G, cov, err=err, genome_seed=k, read_seed=k + 3)
    codes = np.concatenate([m1, m2])
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(codes))
    hc = DistHostCheck(k, counters, insert_batch=insert_batch, claim_log2=12, p2_first=p2_first)
    hc.attach()
    if shared:
        n = codes.shape[0]
        a, b = n * rank // world, n * (rank + 1) // world
        if rank == world - 1 and world > 2:
            a = b  # a rank without reads of its own
        elif rank == world - 2 and world > 2:
            b = n
        words, woff, lens = pack(codes[a:b])
        gw, go, gl, nt = hc.share(words, woff, lens)
        assert nt == n
        hc.load_packed(gw, go, gl, nt)
        cnt = hc.counters()
        rh, ch = hc.assemble_packed(gw, go, gl, nt)
    else:
        hc.load(buf, off)
        cnt = hc.counters()
        rh, ch = hc.assemble(buf, off)
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    ok = {
        "counting_filter": bool(np.array_equal(o.counters(), cnt)),
        "saturated": int(cnt.max()),
        "results": bool(np.array_equal(ro, rh)),
        "contigs": [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch],
        "visited": bool(np.array_equal(o.visited(), hc.visited())),
        "assembly_counters": o.assembly_counters() == hc.assembly_counters(),
        "n_contigs": len(co),
    }
    return ok, hc


def case_kept(rank, world):
    """Reads kept in every rank's store between the passes (abg_keep_reads / abg_load_seqs_v /
    abg_assemble_kept) in a partitioned run: every rank loads every read (several buffers, two calls,
    reads with N and short ones among them) and assembles from its store."""
    k = 37
    m1, m2 = synth.make_read_set(11000, 25.0, err=0.005, genome_seed=k, read_seed=k + 3)
    reads = [bytes(r) for r in synth.codes_to_ascii(np.concatenate([m1, m2]))]
    reads[5] = reads[5][:40] + b"N" + reads[5][41:]
    reads[77] = reads[77][:20]
    reads[300] = reads[300].lower()
    buf, off = api.concat_seqs(reads)
    cuts = [0, 1, 400, 401, 1500, len(reads)]
    chunks = [api.concat_seqs(reads[a:b]) for a, b in zip(cuts, cuts[1:])]
    hc = DistHostCheck(k, 1 << 20, insert_batch=15000, claim_log2=12, p2_first=64)
    hc.attach()
    assert hc.keep_reads(True, len(buf)) == 0
    hc.load_chunks(chunks[:3])
    hc.load_chunks(chunks[3:])
    cnt = hc.counters()
    rc, rh, ch = hc.assemble_kept(len(reads))
    assert rc == 0
    o = ob.Oracle(k, counters=1 << 20)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    ok = {
        "counting_filter": bool(np.array_equal(o.counters(), cnt)),
        "results": bool(np.array_equal(ro, rh)),
        "contigs": [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch],
        "visited": bool(np.array_equal(o.visited(), hc.visited())),
        "assembly_counters": o.assembly_counters() == hc.assembly_counters(),
        "n_contigs": len(co),
    }
    return ok, hc


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    what = sys.argv[1]
    if what == "selftest":
        # abyss_amd.dist.selftest (what bench.py runs on the RCCL communicator before the timed steps of an N > 1 run) over gloo:
        # a sound communicator passes every check on every rank; one whose all_to_all_v delivers a part to the wrong place, or
        # whose all_reduce forgets a rank, is caught
        comm = adist.StagedTorchComm(*adist.host_memory_io())
        good = adist.selftest(comm, adist.HostBuf)
        real_a2a, real_ar = comm._a2a, comm._ar  # (the thunks themselves: a field read back from the struct aliases the field)

        def bad_a2a(user, send, sc, sd, recv, rc, rd, stream):
            rc_ = real_a2a(user, send, sc, sd, recv, rc, rd, stream)
            n = sum(int(rc[q]) for q in range(world))
            if n > 1:
                C.memmove(recv, recv + 1, n - 1)  # (everything one byte early)
            return rc_

        def bad_ar(user, buf, count, dtype, op, stream):
            return 0  # (nothing combined)
        keep = (adist.A2A_FN(bad_a2a), adist.AR_FN(bad_ar))
        comm.struct.all_to_all_v = keep[0]
        bad1 = adist.selftest(comm, adist.HostBuf)
        comm.struct.all_to_all_v = real_a2a
        comm.struct.all_reduce = keep[1]
        bad2 = adist.selftest(comm, adist.HostBuf)
        ok = {"good": good, "bad_a2a": bad1, "bad_all_reduce": bad2}
        box = [None] * world
        dist.all_gather_object(box, ok)
        if rank == 0:
            print("RESULT " + json.dumps({"ranks": box}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if what == "golden":
        ok, hc = case_golden(sys.argv[2], rank, world)
    elif what == "oracle":
        ok, hc = case_oracle(33, 12000, 1 << 20, 25.0, 0.005, 20000, rank, world, shared=False)
    elif what == "bigbatch":
        # every read in ONE batch of PASS 2: the commit has to order thousands of candidates whose
        # contigs overlap, over several passes of its fixed point (the ranks combine their bits' verdicts)
        ok, hc = case_oracle(31, 20000, 1 << 21, 30.0, 0.01, 20000, rank, world, shared=False, p2_first=1 << 20)
    elif what == "tiny_filter":
        # a filter so small that counters saturate and every op conflicts with many others: long
        # reservation chains, several rounds per batch, the distributed hand-over to the drain kernel
        ok, hc = case_oracle(25, 6000, 1 << 13, 40.0, 0.02, 5000, rank, world, shared=False)
    elif what == "saturate":
        # PASS 1 only: 300 copies of one read saturate counters at 255 (CountingBloomFilter.hpp:146-149),
        # homopolymers give runs of identical k-mers; batches of 1000 ops, 3 ranks
        reads = [b"ACGTTGCATGCCGATAGCTAGGATCCATGCAAATTTGGCC"] * 300 + [b"A" * 60, b"T" * 60, b"ACAC" * 20]
        buf, off = api.concat_seqs(reads)
        o = ob.Oracle(21, counters=4096)
        hc = DistHostCheck(21, 4096, insert_batch=1000, claim_log2=8)
        hc.attach()
        o.load(buf, off)
        hc.load(buf, off)
        a, b = o.counters(), hc.counters()
        ok = {"counting_filter": bool(np.array_equal(a, b)), "saturated": int(b.max())}
    elif what == "saturate_tiled":
        # the same through the ranks' tiles: one k-mer hundreds of times in a batch, counters driven to 255
        reads = [b"ACGTTGCATGCCGATAGCTAGGATCCATGCAAGCTTGGCATTCGGATACCGGTAAGCTAGCTAACGGT"] * 400 + [b"A" * 150] * 12 + [b"AC" * 75] * 8
        buf, off = api.concat_seqs(reads)
        o = ob.Oracle(40, counters=1 << 22)
        hc = DistHostCheck(40, 1 << 22, insert_batch=30000, claim_log2=16)
        hc.attach()
        o.load(buf, off)
        hc.load(buf, off)
        a, b = o.counters(), hc.counters()
        ok = {"counting_filter": bool(np.array_equal(a, b)), "saturated": int(b.max())}
    elif what == "kept":
        ok, hc = case_kept(rank, world)
    elif what == "shared":
        ok, hc = case_oracle(41, 10000, 1 << 19, 25.0, 0.01, 15000, rank, world, shared=True)
    elif what == "sliced_checkpoint":
        # a sliced filter written out rank by rank (abg_counters_export) and read back into a fresh sliced context
        # (abg_counters_import: each rank takes its own range of the host copy), which then runs PASS 2
        k, counters = 41, 1 << 19
        m1, m2 = synth.make_read_set(10000, 25.0, err=0.01, genome_seed=k, read_seed=k + 3)
        buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
        a = DistHostCheck(k, counters, insert_batch=15000, claim_log2=12, p2_first=64)
        a.attach()
        a.load(buf, off)
        saved = a.counters()
        hc = DistHostCheck(k, counters, insert_batch=15000, claim_log2=12, p2_first=64)
        hc.attach()
        hc.l.hc_counters_import.argtypes = [C.c_void_p, C.c_void_p]
        assert hc.l.hc_counters_import(hc.h, saved.ctypes.data) == 0
        cnt = hc.counters()
        rh, ch = hc.assemble(buf, off)
        o = ob.Oracle(k, counters=counters)
        o.load(buf, off)
        ro, co = o.assemble(buf, off)
        ok = {
            "counting_filter": bool(np.array_equal(o.counters(), cnt)) and bool(np.array_equal(saved, cnt)),
            "results": bool(np.array_equal(ro, rh)),
            "contigs": [contig_tuple(c) for c in co] == [contig_tuple(c) for c in ch],
            "visited": bool(np.array_equal(o.visited(), hc.visited())),
            "assembly_counters": o.assembly_counters() == hc.assembly_counters(),
            "n_contigs": len(co), "held_fraction": hc.stats()["counter_bytes_held"] / float(counters),
        }
        del a
    elif what == "sliced":
        # B beyond one device (ABG_SLICE_FILTER=1, set by the test): every rank holds its own range of the counters and nothing
        # else -- under tests/hostcheck the rest of the array is address space without memory, so a stray access kills the rank --
        # PASS 2 probes the gathered bit plane, coverage comes through an all-reduce
        ok, hc = case_oracle(41, 10000, 1 << 19, 25.0, 0.01, 15000, rank, world, shared=(len(sys.argv) > 2 and sys.argv[2] == "shared"))
        held = hc.stats()["counter_bytes_held"]
        ok["held_fraction"] = held / float(1 << 19)
        ok["direct_access_refused"] = not bool(hc.l.hc_counters(hc.h))
    else:
        raise SystemExit("unknown case")
    # (the walkers' own work counters are per rank: each rank walks its share of the candidates)
    ok["stats"] = {k: v for k, v in hc.stats().items() if k not in ("bulk_calls", "bulk_steps", "lin_steps", "chain_steps", "memo_hits", "memo_adds", "cls_covered_reads", "archive_bases")}
    ok["comm_calls"] = hc.comm.calls
    # every rank must have reached the same verdicts
    flat = json.dumps({k: v for k, v in ok.items() if k not in ("comm_calls", "comm_pass1")}, sort_keys=True)  # (what a rank sent is its own business)
    box = [None] * world
    dist.all_gather_object(box, flat)
    ok["ranks_agree"] = all(b == box[0] for b in box)
    if rank == 0:
        print("RESULT " + json.dumps(ok), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
