"""Worker of tests/test_gpu_dist.py: one rank of a partitioned run on a GPU box with a single
GPU.  All ranks share cuda:0 (RCCL refuses duplicate devices, so the ranks are joined by gloo
through host copies -- abyss_amd.dist.StagedTorchComm over abg_dev_copy); everything else is the
product path: libabyss_amd.so kernels on device memory.  Rank 0 compares with the oracle.
TEST INFRASTRUCTURE: launched with torch.distributed.run."""
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
from abyss_amd import api, dist as adist, synth  # noqa: E402
from util import contig_tuple  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    k, counters = 48, 1 << 22
    m1, m2 = synth.make_read_set(40000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    g = api.BloomDBG(k, counters=counters, device=0, insert_batch_kmers=1 << 18, claim_log2=22)
    comm = adist.StagedTorchComm(*adist.device_memory_io(g))
    g.attach_comm(comm)
    g.load(buf, off)
    cnt = g.counters()
    rg, cg = g.assemble(buf, off)
    ok = {}
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    ro, co = o.assemble(buf, off)
    ok["counting_filter"] = bool(np.array_equal(o.counters(), cnt))
    ok["results"] = bool(np.array_equal(ro, rg))
    ok["contigs"] = [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    ok["visited"] = bool(np.array_equal(o.visited(), g.visited()))
    ok["assembly_counters"] = o.assembly_counters() == g.assembly_counters()
    ok["n_contigs"] = len(co)
    ok["stats"] = g.stats()
    box = [None] * world
    dist.all_gather_object(box, json.dumps(ok, sort_keys=True))
    ok["ranks_agree"] = all(b == box[0] for b in box)
    ok["comm_calls"] = comm.calls
    if rank == 0:
        print("RESULT " + json.dumps(ok), flush=True)
    g.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
