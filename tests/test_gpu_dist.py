"""The partitioned multi-GPU path on a real MI355X (a box with ONE GPU), in worker processes
(tests/dist_gpu_worker.py) that load torch first like bench.py does:

* the library's RCCL communicator with a single rank and ABG_FORCE_DIST=1 -- the partitioned
  kernels (FHashClaimT<true>, FEvalDist / FApplyDist, the hipcub compaction, the drain hand-over,
  the split classification and the merge of walk results) and every RCCL call (ncclAllReduce,
  in-place ncclAllGather) run on the device, each collective an identity;
* two and three ranks sharing the GPU, joined by gloo through host copies.

Both must reproduce the reference's single sequential run bit for bit, and exit cleanly."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def result_of(r):
    assert r.returncode == 0, (r.returncode, r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    return json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("RESULT ")][-1][7:])


def test_rccl_single_rank_partitioned_path_reproduces_reference_runs_and_oracle():
    env = dict(os.environ, ABG_FORCE_DIST="1")
    r = subprocess.run([sys.executable, WORKER, "rccl1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("k64", "k40_mixed", "k48_K16", "share_total", "counting_filter", "results", "contigs", "visited",
                "assembly_counters"):
        assert out[key], (key, out)
    assert out["n_contigs"] > 10
    assert all(v > 0 for v in out["launches"].values()), out["launches"]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_match_oracle(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.pop("ABG_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), WORKER, "staged"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("counting_filter", "results", "contigs", "visited", "assembly_counters", "ranks_agree"):
        assert out[key], (key, out)
    assert out["n_contigs"] > 10
    assert out["comm_calls"]["all_reduce"] > 0
