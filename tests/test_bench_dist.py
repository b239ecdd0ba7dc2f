"""The N>1 path of bench.py (one process per GPU): the whole-job aggregation (MAX of the timed region, SUM of units) under torch.distributed with
world_size 2 on the gloo backend, and the JSON contract of the bench line."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_aggregate_world2_gloo(tmp_path):
    script = tmp_path / "agg.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import torch.distributed as dist
        import bench
        dist.init_process_group(backend="gloo")
        rank = dist.get_rank()
        # rank 0 took 2.0 s for 100 units/step, rank 1 took 3.5 s for 150 units/step, 2 steps each
        t, n = bench.aggregate(2.0 if rank == 0 else 3.5, 100 if rank == 0 else 150, 2, dist.get_world_size())
        if rank == 0:
            print(json.dumps({"t": t, "n": n}))
        dist.barrier()
        dist.destroy_process_group()
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"t": 3.5, "n": 500}


def test_aggregate_single_rank():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(1.25, 87, 3, 1) == (1.25, 261)


def test_gpus_flag_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 8` (no torchrun around it) must not time one GPU and call it eight."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    assert bench.main() == 0
    (cmd, env), = calls
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_evidence_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py quotes TCC traffic from profiles/*_pmc_traffic.json: the file taken on exactly these kernel sources ("_csrc_sha256" ==
    csrc_digest()) if there is one, else the newest -- and says which (roofline.traffic_kernels_are_head)."""
    sys.path.insert(0, ROOT)
    import bench
    (tmp_path / "abyss_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (tmp_path / "profiles").mkdir()
    (tmp_path / "abyss_amd" / "csrc" / "a.h").write_text("int a;\n")
    (tmp_path / "include" / "abyss_amd.h").write_text("int b;\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    d1 = bench.csrc_digest()
    (tmp_path / "profiles" / "r01_pmc_traffic.json").write_text(json.dumps({"FWalk": {}, "_csrc_sha256": d1}))
    (tmp_path / "profiles" / "r02_pmc_traffic.json").write_text(json.dumps({"FWalk": {}, "_csrc_sha256": "somethingelse"}))
    path, j, head, behind = bench.pick_evidence("_pmc_traffic.json")
    assert path.endswith("r01_pmc_traffic.json") and head and behind == 0
    (tmp_path / "abyss_amd" / "csrc" / "a.h").write_text("int a2;\n")  # a kernel changes: neither file is the head's any more
    assert bench.csrc_digest() != d1
    path, j, head, behind = bench.pick_evidence("_pmc_traffic.json")
    assert path.endswith("r02_pmc_traffic.json") and not head and behind is None
    assert bench.pick_evidence("_nothing.json") == (None, None, False, None)
