"""The partitioned multi-GPU path on a real MI355X (a box with ONE GPU):

* the library's RCCL communicator with a single rank and ABG_FORCE_DIST=1 -- the partitioned
  kernels (FHashClaimT<true>, FEvalDist / FApplyDist, the hipcub compaction, the drain hand-over,
  the split classification and the merge of walk results) and every RCCL call (ncclAllReduce,
  in-place ncclAllGather, grouped ncclBroadcast) run on the device, each collective an identity;
* two and three ranks sharing the GPU, joined by gloo through host copies.

Both must reproduce the reference's single sequential run bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api, dist as adist, synth
from util import GoldenCase, contig_tuple, mask_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def force_dist(monkeypatch):
    monkeypatch.setenv("ABG_FORCE_DIST", "1")


@pytest.mark.parametrize("name", ["k64", "k40_mixed", "k48_K16"])
def test_rccl_single_rank_partitioned_path_reproduces_reference_run(name, force_dist):
    g0 = GoldenCase(name)
    kw = g0.kwargs()
    g = api.BloomDBG(kw["k"], counters=g0.meta["counters"], num_hashes=kw["num_hashes"], min_cov=kw["min_cov"],
                     trim=kw["trim"], spaced_seed=mask_of(g0), insert_batch_kmers=1 << 16, claim_log2=22)
    comm = adist.RcclComm(0, single=True)
    g.attach_comm(comm)
    g.load(g0.buf, g0.off)
    assert g.counting_stats()[1] == g0.meta["filtered_popcount"]
    results, contigs = g.assemble(g0.buf, g0.off)
    assert api.format_fasta(contigs, g0.ids) == g0.fasta
    assert api.format_read_log(results, g0.ids) == g0.readlog
    assert api.format_trace(contigs, g0.ids, g0.reads, g0.opts["k"], with_length=False) == g0.trace
    g.close()
    comm.close()


def test_rccl_single_rank_share_reads_and_oracle(force_dist):
    """abg_share_reads + packed entry points on the partitioned path, against the oracle."""
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
    assert nt == n
    g.load_packed(gw, go, gl, nt)
    o = ob.Oracle(k, counters=counters)
    o.load(buf, off)
    assert np.array_equal(o.counters(), g.counters())
    rg, cg = g.assemble_packed(gw, go, gl, nt)
    ro, co = o.assemble(buf, off)
    assert np.array_equal(ro, rg)
    assert [contig_tuple(c) for c in co] == [contig_tuple(c) for c in cg]
    assert np.array_equal(o.visited(), g.visited())
    assert o.assembly_counters() == g.assembly_counters()
    assert g.profile_get("comm_all_reduce")[1] > 0 and g.profile_get("comm_all_gather")[1] > 0
    assert g.profile_get("compact")[1] > 0 and g.profile_get("insert_apply")[1] > 0
    g.close()
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_oracle(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.pop("ABG_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + world),
                        os.path.join(ROOT, "tests", "dist_gpu_worker.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("RESULT ")][-1][7:])
    for key in ("counting_filter", "results", "contigs", "visited", "assembly_counters", "ranks_agree"):
        assert out[key], (key, out)
    assert out["n_contigs"] > 10
    assert out["comm_calls"]["all_reduce"] > 0
