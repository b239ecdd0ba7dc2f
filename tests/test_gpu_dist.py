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
    # (round 6: the single-GPU rules on the partitioned path leave this small run's rounds nothing -- the partitioned reservation
    # rounds are exercised by the same run under round 2's rule below)
    assert all(v > 0 for nm, v in out["launches"].items() if nm != "insert_apply"), out["launches"]
    assert out["launches"]["dist_pack"] > 0 and out["launches"]["co_settle"] > 0, out["launches"]
    env["ABG_COSETTLE"] = "0"
    r = subprocess.run([sys.executable, WORKER, "rccl1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("k64", "k40_mixed", "k48_K16", "counting_filter", "results", "contigs", "visited", "assembly_counters"):
        assert out[key], (key, out)
    assert out["launches"]["insert_apply"] > 0 and out["launches"]["co_settle"] == 0, out["launches"]


def test_rccl_single_rank_routed_path_reproduces_reference_runs_and_oracle():
    """The routed form of PASS 1 (Engine::insert_tiles_routed) with the real kernels -- FRoutePack, FBinCoarseRec,
    FRouteReply / FRouteCombine / FRouteTgt, the pending records -- on one rank, each exchange a device copy."""
    env = dict(os.environ, ABG_FORCE_DIST="1", ABG_DIST_ROUTE_MIN="1")
    r = subprocess.run([sys.executable, WORKER, "rccl1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("k64", "k40_mixed", "k48_K16", "share_total", "counting_filter", "results", "contigs", "visited",
                "assembly_counters"):
        assert out[key], (key, out)
    assert out["launches"]["route_pack"] > 0 and out["launches"]["route_reply"] > 0, out["launches"]


@pytest.mark.parametrize("world,route", [(2, "0"), (3, "0"), (2, "2"), (3, "2")])
def test_ranks_sharing_one_gpu_match_oracle(world, route):
    """route "2": the (op, counter) pairs are routed to the ranks that own the counters (abg_comm::all_to_all_v, here
    through gloo on host copies); "0": the all-gather form."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", ABG_DIST_ROUTE_MIN=route)
    env.pop("ABG_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), WORKER, "staged"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("counting_filter", "results", "contigs", "visited", "assembly_counters", "ranks_agree"):
        assert out[key], (key, out)
    assert out["n_contigs"] > 10
    assert out["comm_calls"]["all_reduce"] > 0
    assert (out["comm_calls"].get("all_to_all_v", 0) > 0) == (route != "0"), out["comm_calls"]


@pytest.mark.parametrize("world,route", [(2, "0"), (3, "2")])
def test_sliced_filter_on_ranks_sharing_one_gpu(world, route):
    """abg_params.slice_filter (ABG_SLICE_FILTER=1): each rank allocates its own range of the counters only, PASS 2 probes the
    gathered bit plane, coverage goes through FPcCover and an all-reduce -- the real kernels, against the oracle (the counting
    filter is exported rank by rank).  The CPU twin of this test runs with the other ranks' counters unmapped
    (tests/test_dist_partition.py::test_filter_that_fits_no_single_rank)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", ABG_DIST_ROUTE_MIN=route, ABG_SLICE_FILTER="1")
    env.pop("ABG_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29710 + world), WORKER, "staged"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("counting_filter", "results", "contigs", "visited", "assembly_counters", "ranks_agree"):
        assert out[key], (key, out)
    assert out["n_contigs"] > 10
    assert 0 < out["stats"]["counter_bytes_held"] <= (1 << 22) // world + 128, out["stats"]


def test_rccl_single_rank_sliced_filter_reproduces_reference_runs_and_oracle():
    """The sliced filter through the library's RCCL communicator (one rank): the plane's all-gather, the coverage all-reduce and
    the rank-by-rank export are RCCL calls on the engine's stream."""
    env = dict(os.environ, ABG_FORCE_DIST="1", ABG_SLICE_FILTER="1")
    r = subprocess.run([sys.executable, WORKER, "rccl1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = result_of(r)
    for key in ("k64", "k40_mixed", "k48_K16", "share_total", "counting_filter", "results", "contigs", "visited",
                "assembly_counters"):
        assert out[key], (key, out)
    assert out["launches"]["pc_cover"] > 0 and out["launches"]["solid_plane"] > 0 and out["held"] == (1 << 24) + 64, out


def test_bench_runs_configs4_scaled_down_on_two_ranks():
    """`bench.py --gpus N --config 4` (k=96, B=500G, 1.2 G pairs on eight GPUs) scaled down to what two ranks sharing this
    GPU can hold: a sliced filter, the reads in three chunks per pass (every chunk all-gathered), the launch path of the
    driver's multi-GPU run (ranks started by bench.py itself)."""
    env = dict(os.environ, ABG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("ABG_FORCE_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "staged", "--slice-filter", "--chunks", "3",
                        "--config", "4", "--pairs", "300000", "--bloom", "120M", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-end-to-end"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_agree"] is True and c["chunks"] == 3 and "k=96" in c["workload"], c
    assert 0 < c["counter_bytes_per_rank"] < 0.51 * (120 << 20) / 1.125 + 4096, c
    assert c["unitigs"] > 1000 and d["value"] > 0


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: the first real multi-rank run of the RCCL communicator")
def test_two_real_ranks_over_rccl_agree_with_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (one process per GPU, backend nccl = RCCL): the
    library's own communicator (in-place ncclAllGather, the ragged broadcast group, all_reduce per
    round) between two real ranks, against the same job on one GPU; then the host binary's --gpus 2."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ABG_FORCE_DIST", None)
    args = ["--pairs", "400000", "--bloom", "160M", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=env, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    a = json.loads([ln for ln in one.stdout.decode().splitlines() if ln.startswith("{")][-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--scaling", "strong"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    b = json.loads([ln for ln in two.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["config"]["ranks_agree"] is True and "note" not in b["config"], b["config"]
    # (the two runs draw their reads with different seeds -- every rank generates its own share -- so the
    # unitig sets are those of two samples of the same genome: close, not equal)
    assert abs(b["config"]["unitig_bp"] - a["config"]["unitig_bp"]) < 0.02 * a["config"]["unitig_bp"]
    # the drop-in binary with --gpus 2 writes the FASTA of the one-GPU run
    from abyss_amd import build, synth
    m1, m2 = synth.make_read_set(200000, 40.0)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)
    cli = build.build_cli()
    r1 = subprocess.run([cli, "-k32", "-q3", "-b100M", "r1.fq", "r2.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    r2 = subprocess.run([cli, "-k32", "-q3", "-b100M", "--gpus=2", "r1.fq", "r2.fq"], cwd=tmp_path, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0 and r2.returncode == 0, r2.stderr.decode()[-2000:]
    assert r1.stdout == r2.stdout and len(r1.stdout) > 100000
