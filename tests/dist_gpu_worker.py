"""Worker of tests/test_gpu_dist.py (TEST INFRASTRUCTURE), run as a separate process so that torch
-- and with it the HIP runtime and RCCL it bundles -- is loaded before libabyss_amd.so, exactly
as in bench.py (one HIP runtime, one RCCL per process).

  staged   one rank of a partitioned run on a box with a single GPU: all ranks share cuda:0 (RCCL
           refuses duplicate devices, so they are joined by gloo through host copies,
           abyss_amd.dist.StagedTorchComm over abg_dev_copy); launched with torch.distributed.run
  rccl1    the library's RCCL communicator with ONE rank and ABG_FORCE_DIST=1: every partitioned
           kernel, the compaction, the merges and each RCCL call run on the device, each collective
           an identity; golden reference runs + abg_share_reads against the oracle
Rank 0 prints one "RESULT {json}" line."""
import json
import os
import sys

import torch  # noqa: F401  (first: see above)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
from abyss_amd import api, dist as adist, synth  # noqa: E402
from util import GoldenCase, contig_tuple, mask_of  # noqa: E402


def against_oracle(g, k, counters, buf, off, packed=None):
    """PASS 1 + PASS 2 of context g against the oracle; packed = (gw, go, gl, n) to use the packed entry points."""
    ok = {}
    if packed:
        g.load_packed(*packed)
    else:
        g.load(buf, off)
    cnt = g.counters()
    rg, cg = g.assemble_packed(*packed) if packed else g.assemble(buf, off)
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    ok["counting_filter"] = bool(np.array_equal(o.counters(), cnt))
    ok["results"] = bool(np.array_equal(ro, rg))
    ok["contigs"] = [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    ok["visited"] = bool(np.array_equal(o.visited(), g.visited()))
    ok["assembly_counters"] = o.assembly_counters() == g.assembly_counters()
    ok["n_contigs"] = len(co)
    return ok


def staged():
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    k, counters = 48, 1 << 22
    m1, m2 = synth.make_read_set(40000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    g = api.BloomDBG(k, counters=counters, device=0, insert_batch_kmers=1 << 18, claim_log2=22)
    comm = adist.StagedTorchComm(*adist.device_memory_io(g))
    g.attach_comm(comm)
    ok = against_oracle(g, k, counters, buf, off)
    ok["stats"] = {k: v for k, v in g.stats().items() if k not in ("bulk_calls", "bulk_steps", "lin_steps", "chain_steps", "memo_hits", "memo_adds", "cls_covered_reads", "archive_bases")}  # (per-rank work)
    box = [None] * world
    dist.all_gather_object(box, json.dumps(ok, sort_keys=True))
    ok["ranks_agree"] = all(b == box[0] for b in box)
    ok["comm_calls"] = comm.calls
    if rank == 0:
        print("RESULT " + json.dumps(ok), flush=True)
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def rccl1():
    assert os.environ.get("ABG_FORCE_DIST") == "1"
    ok = {}
    for name in ("k64", "k40_mixed", "k48_K16"):
        g0 = GoldenCase(name)
        kw = g0.kwargs()
        g = api.BloomDBG(kw["k"], counters=g0.meta["counters"], num_hashes=kw["num_hashes"], min_cov=kw["min_cov"],
                         trim=kw["trim"], spaced_seed=mask_of(g0), insert_batch_kmers=1 << 16, claim_log2=22)
        comm = adist.RcclComm(0, single=True)
        g.attach_comm(comm)
        g.load(g0.buf, g0.off)
        fp = g.counting_stats()[1]
        results, contigs = g.assemble(g0.buf, g0.off)
        ok[name] = (fp == g0.meta["filtered_popcount"] and api.format_fasta(contigs, g0.ids) == g0.fasta
                    and api.format_read_log(results, g0.ids) == g0.readlog
                    and api.format_trace(contigs, g0.ids, g0.reads, g0.opts["k"], with_length=False) == g0.trace)
        g.close()
        comm.close()
    # abg_share_reads + the packed entry points
    k, counters = 64, 1 << 24
    m1, m2 = synth.make_read_set(120000, 30.0)
    codes = np.concatenate([m1, m2])
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(codes))
    n, L = codes.shape
    wpr = (L + 15) // 16
    pad = np.zeros((n, wpr * 16), dtype=np.uint64)
    pad[:, :L] = codes
    words = (pad.reshape(n, wpr, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=2).astype(np.uint32)
    g = api.BloomDBG(k, counters=counters, insert_batch_kmers=1 << 20, claim_log2=24)
    dw = g.to_device(words.reshape(-1))
    do = g.to_device(np.arange(n + 1, dtype=np.uint64) * np.uint64(wpr))
    dl = g.to_device(np.full(n, L, dtype=np.uint32))
    comm = adist.RcclComm(0, single=True)
    g.attach_comm(comm)
    g.profile_enable(True)
    gw, go, gl, nt = g.share_reads(dw, do, dl, n)
    ok["share_total"] = nt == n
    ok.update(against_oracle(g, k, counters, buf, off, packed=(gw, go, gl, nt)))
    ok["launches"] = {nm: g.profile_get(nm)[1] for nm in ("comm_all_reduce", "comm_all_gather", "compact", "insert_apply", "merge_fix", "dist_pack", "co_settle")}
    if os.environ.get("ABG_SLICE_FILTER") == "1":
        ok["launches"].update({nm: g.profile_get(nm)[1] for nm in ("pc_cover", "solid_plane")})
        ok["held"] = g.stats()["counter_bytes_held"]
    if os.environ.get("ABG_DIST_ROUTE_MIN") == "1":
        ok["launches"].update({nm: g.profile_get(nm)[1] for nm in ("route_pack", "route_reply", "route_tgt")})
    g.close()
    comm.close()
    print("RESULT " + json.dumps(ok), flush=True)


if __name__ == "__main__":
    {"staged": staged, "rccl1": rccl1}[sys.argv[1]]()
