#!/usr/bin/env python3
"""bench.py -- Mk-mers/s of the abyss-bloom-dbg unitig stage on MI355X.

One "step" = one whole pass of the hot path over one synthetic read set already resident
in HBM in the packed 2-bit layout: reset both filters, PASS 1 (ntHash + ordered
conservative-update insert of every read k-mer into the counting Bloom filter), PASS 2
(classify reads, walk unitigs, commit contigs in read order).  N = number of read k-mers,
each counted once although both passes touch it.

Workload: one GPU times BASELINE.json configs[1], "E. coli-scale synthetic: 5 M x 2x150 bp reads, k=64,
B=2G, H=4" (30 Mbp genome, 50x, 0.5 % substitution errors); --config 2 / 3 select configs[2]
(200 M pairs, B=40G) / configs[3] (-k96 -K32).  With --gpus N (one process per GPU; run without a
launcher, bench.py starts its N ranks itself) ONE job is split over the ranks -- by default
configs[2], the configuration BASELINE.json lists for the partitioned filter, strong-scaled
("scaling": "strong"; its one-GPU point is replayed from profiles/ in `strong_scaling`): every rank
holds 1/N of the reads, the counting filter is range-partitioned by position over the ranks' HBM
during PASS 1 (RCCL), gathered for PASS 2, whose walks are split over the ranks and merged before the
ordered commit; the unitigs are bit-identical to a 1-GPU run of that job.
--scaling weak makes the job N times --pairs / --bloom instead; --mode replicas runs N independent
copies of the job (no collective on the data path).  --config 4 is configs[4] (1.2 G pairs, k=96, B=500G
on eight GPUs): a filter beyond one GPU -- each rank keeps its own range of the counters, PASS 2 probes
the all-gathered bit plane (abg_params.slice_filter; --slice-filter forces it at any size) -- and each
pass is fed its reads in --chunks all-gathered pieces, so that no rank holds the read set.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from abyss_amd import api, synth  # noqa: E402

METRIC = "Mk-mers/s inserted+extended (abyss-bloom-dbg, k=64); unitig bit-exact"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def gen_packed_reads(genome_len: int, n_pairs: int, read_len: int, err: float, seed: int, device, read_seed: int = 7,
                     first: int = 0, total_pairs: int = None):
    """The synthetic read set, generated on the GPU in the packed layout of include/abyss_amd.h (2 bits
    per base, 16 bases per uint32 word, each read on a word boundary; all mate-1 reads, then all mate-2
    reads: the file order of the reference, BloomIO.h:102-115).  synth.packed_reads_torch is the
    torch twin of synth.make_read_set_cb, whose FASTQ files the reference binary was run on at -j1 for
    the digests under tests/golden/full_size.json: the reads timed here are THAT read set."""
    h1, h2 = synth.make_genome(genome_len, seed=seed)
    return synth.packed_reads_torch(h1, h2, n_pairs, read_len, err, read_seed, device, first=first, total_pairs=total_pairs)


def csrc_digest():
    """sha256 over the kernel sources (abyss_amd/csrc/*.h, *.hip, include/abyss_amd.h): what evidence under profiles/ is
    tied to.  (The GPU boxes get the tree without .git, so a commit id cannot be checked there; the sources can.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "abyss_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "abyss_amd.h")]:
        p = os.path.join(d, f)
        if os.path.isfile(p) and (f.endswith(".h") or f.endswith(".hip")):
            h.update(f.encode() + b"\0" + open(p, "rb").read())
    return h.hexdigest()


def pick_evidence(suffix: str):
    """The file profiles/*<suffix> taken on THESE kernel sources ("_csrc_sha256" equals csrc_digest()), else the newest one by
    name; returns (path, json, matches_head, commits_behind or None).  commits_behind: commits touching abyss_amd/csrc since the
    file's "_commit", where git is there to ask."""
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith(suffix)) if os.path.isdir(pdir) else []
    if not cands:
        return None, None, False, None
    head = csrc_digest()
    loaded = []
    for f in cands:
        try:
            loaded.append((f, json.load(open(os.path.join(pdir, f)))))
        except Exception:  # noqa: BLE001
            pass
    best = [x for x in loaded if x[1].get("_csrc_sha256") == head]
    f, j = (best or loaded)[-1]
    behind = None
    if not best and j.get("_commit") and os.path.isdir(os.path.join(ROOT, ".git")):
        r = subprocess.run(["git", "-C", ROOT, "rev-list", "--count", "%s..HEAD" % j["_commit"], "--", "abyss_amd/csrc"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if r.returncode == 0 and r.stdout.strip().isdigit():
            behind = int(r.stdout.strip())
    return os.path.join("profiles", f), j, bool(best), (0 if best else behind)


def golden_for(config: int, pairs: int, k: int, K: int, bloom: str):
    """The reference's -j1 result on this very read set (tests/golden/full_size.json), or None."""
    src = os.path.join(ROOT, "tests", "golden", "full_size.json")
    if not os.path.exists(src):
        return None
    for g in json.load(open(src)).get("runs", []):
        if g.get("generator") == "make_read_set_cb" and (g["pairs"], g["k"], g.get("K", 0), g["bloom"]) == (pairs, k, K, bloom):
            return g
    return None


def end_to_end(a, genome_len, read_len, err, device, golden):
    """The drop-in binary on the full read set as FASTQ files: process start to last unitig written
    (parse, pack, upload, both passes, FASTA out), and the FASTA's sha256 against the reference's."""
    import hashlib
    from abyss_amd import build
    h1, h2 = synth.make_genome(genome_len, seed=42)
    with tempfile.TemporaryDirectory() as td:
        t0 = time.time()
        m1, m2 = synth.sample_pairs_cb(h1, h2, a.pairs, read_len=read_len, err=err, seed=7)
        synth.write_fastq(os.path.join(td, "r1.fq"), m1, "r", 1)
        synth.write_fastq(os.path.join(td, "r2.fq"), m2, "r", 2)
        prep = time.time() - t0
        del m1, m2
        args = [build.build_cli(), "-k%d" % a.k, "-b%s" % a.bloom, "-H4", "-q3", "-j%d" % (os.cpu_count() or 1)]
        if a.K:
            args.append("-K%d" % a.K)
        # the files just written are 3 GB of dirty pages: let the kernel write them back before anything is timed (they stay in the
        # page cache), and take three runs (the median is reported, the list kept) -- on a box whose host is busy a single run has measured anything between
        # 1.3 and 4.7 s for the same 0.63 s of device work; every run's wall time is in wall_ms_runs
        try:
            os.sync()
        except Exception:
            pass
        # (a GPU process that starts within a second of another one's exit waits ~0.8 s in its first large hipMalloc while the driver
        # scrubs the ~27 GB the other one freed -- measured: [1173, 1978, 2039], [1222, 1981, 2007] ms for runs back to back, the first
        # always the fast one.  A user runs the binary once: the runs start ABG_E2E_GAP_S (3 s) after the GPU was last let go of.)
        gap = float(os.environ.get("ABG_E2E_GAP_S", "3"))
        walls, r = [], None
        for _ in range(3):
            time.sleep(gap)
            t0 = time.time()
            ri = subprocess.run(args + ["r1.fq", "r2.fq"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            walls.append(time.time() - t0)
            if r is None or ri.returncode != 0 or walls[-1] == min(walls):
                r = ri
            if ri.returncode != 0:
                break
        wall = sorted(walls)[len(walls) // 2] if len(walls) == 3 else min(walls)  # the median of three
        adj = rr = None
        if r.returncode == 0 and golden and "adjlist" in golden:
            # abyss-pe's next step on the unitigs just written (AdjList $(alopt) --dot, bin/abyss-pe:575-577)
            open(os.path.join(td, "unitigs-1.fa"), "wb").write(r.stdout)
            time.sleep(gap)
            t0 = time.time()
            ra = subprocess.run([os.path.join(build.BIN_DIR, "AdjList")] + golden["adjlist"]["options"].split() + ["unitigs-1.fa"],
                                cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ABG_ADJ_TIMING="1"))
            adj = {"what": "abyss_amd/bin/AdjList %s on those unitigs, process start to graph written" % golden["adjlist"]["options"],
                   "rc": ra.returncode, "wall_ms": round((time.time() - t0) * 1e3), "edges": ra.stdout.count(b" -> "),
                   "dot_sha256": hashlib.sha256(ra.stdout).hexdigest(),
                   "matches_reference_dot": bool(hashlib.sha256(ra.stdout).hexdigest() == golden["adjlist"]["dot_sha256"]),
                   "kernels_ms": {l.split()[1]: float(l.split()[2]) for l in ra.stderr.decode().splitlines() if l.startswith("[timing]")}}
            if ra.returncode == 0 and "rresolver" in golden:
                # ... and the rule after that in Bloom mode (abyss-rresolver-short, bin/abyss-pe:581-585): the reads once more, their
                # r-mers into a Bloom filter on the GPU, the repeats' paths tested against it
                open(os.path.join(td, "unitigs-1.dot"), "wb").write(ra.stdout)
                cmd = [os.path.join(build.BIN_DIR, "abyss-rresolver-short"), "-b" + a.bloom, "-f0.8", "-j%d" % (os.cpu_count() or 1), "-k%d" % a.k,
                       "-h", "rr", "--dot", "-c", "rr.fa", "-g", "rr.dot", "unitigs-1.fa", "unitigs-1.dot", "r1.fq", "r2.fq"]
                rwalls, rq = [], None
                for _ in range(3):
                    time.sleep(gap)  # (as above: not within a second of another GPU process's exit)
                    t0 = time.time()
                    rq = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ABG_RR_TIMING="1"))
                    rwalls.append(time.time() - t0)
                    if rq.returncode != 0:
                        break
                files = {f: hashlib.sha256(open(os.path.join(td, f), "rb").read()).hexdigest() for f in sorted(os.listdir(td)) if f.startswith("rr")}
                kern = {}
                for l in rq.stderr.decode().splitlines():
                    if l.startswith("[timing]"):
                        kern[l.split()[1]] = kern.get(l.split()[1], 0.0) + float(l.split()[2])
                ref = golden["rresolver"]
                rr = {"what": "abyss_amd/bin/abyss-rresolver-short %s on those unitigs, that graph and the FASTQ files, process start to the resolved "
                              "contigs and graph written" % " ".join(cmd[1:5]),
                      "rc": rq.returncode, "wall_ms": round(sorted(rwalls)[len(rwalls) // 2] * 1e3), "wall_ms_runs": [round(w * 1e3) for w in rwalls],
                      "wall_ms_is": "the median of the runs listed", "contigs": open(os.path.join(td, "rr.fa"), "rb").read().count(b">") if rq.returncode == 0 else None,
                      "sha256": files, "matches_reference": bool(files == ref["sha256"]), "kernels_ms": kern,
                      "reference_wall_s": ref.get("reference_wall_s"),
                      "reference_is": "oracle/_ref/abyss-rresolver-short (unmodified RResolver sources over oracle/shim/btllib; parity with a real btllib unpinned), "
                                      "timed once by tests/golden/make_full_size_rr.py"}
    kmers = 2 * a.pairs * (read_len - a.k + 1)
    out = {"what": "abyss_amd/bin/abyss-bloom-dbg on the FASTQ files of this read set (two %.2f GB files, page cache warm), process "
                   "start to last unitig written" % (a.pairs * (2 * read_len + 12) / 1e9),
           "rc": r.returncode, "wall_ms": round(wall * 1e3), "wall_ms_runs": [round(w * 1e3) for w in walls],
           "wall_ms_is": "the median of the runs listed", "wall_ms_best": round(min(walls) * 1e3), "value": kmers / wall / 1e6, "unit": "Mk-mers/s",
           "threads": os.cpu_count() or 1, "unitigs": r.stdout.count(b">"), "fasta_sha256": hashlib.sha256(r.stdout).hexdigest(),
           "files_written_in_s": round(prep, 1), "measured": "in this run"}
    if golden:
        out["matches_reference_fasta"] = bool(out["fasta_sha256"] == golden["fasta_sha256"])
    if adj:
        out["adjlist"] = adj
    if rr:
        out["rresolver"] = rr
    return out


def cpu_baseline(k: int, cores: int, K: int = 0, target_s: float = 20.0):
    """The unmodified reference binary (oracle/_ref, kind "reference") -- or the oracle's C port
    when it is not built -- timed on the host cores on a bounded sample of the same workload
    (same read length, coverage, error rate, k, H; genome and B scaled down together)."""
    import oracle_binding as ob
    genome, cov, L = 2_400_000, 50.0, 150  # ~70 M read k-mers: 15-20 s on the 256-thread host
    m1, m2 = synth.make_read_set(genome, cov, read_len=L)
    n_reads = 2 * m1.shape[0]
    kmers = n_reads * (L - k + 1)
    kopt = ["-k%d" % k] + (["-K%d" % K] if K else [])
    sample = "%d x 2x%d bp reads of a %d bp genome (50x, 0.5%% err), k=%d%s, B=160M, H=4" % (m1.shape[0], L, genome, k, " -K%d" % K if K else "")
    if ob.have_ref():
        with tempfile.TemporaryDirectory() as td:
            synth.write_fastq(os.path.join(td, "r1.fq"), m1, "r", 1)
            synth.write_fastq(os.path.join(td, "r2.fq"), m2, "r", 2)
            # fixed start-up cost of the reference (contigEndKmers.rehash(2^28), bloom-dbg.h:993)
            open(os.path.join(td, "one.fq"), "w").write("@x\n%s\n+\n%s\n" % ("A" * L, "I" * L))
            t0 = time.time()
            ob.run_ref(kopt + ["-b1M", "one.fq"], cwd=td, threads=1)
            startup = time.time() - t0
            t0 = time.time()
            out, _ = ob.run_ref(kopt + ["-b160M", "-H4", "r1.fq", "r2.fq"], cwd=td, threads=cores)
            wall = time.time() - t0
            # one thread (the deterministic order the GPU path reproduces) on an eighth of the sample
            n8 = m1.shape[0] // 8
            synth.write_fastq(os.path.join(td, "s1.fq"), m1[:n8], "r", 1)
            synth.write_fastq(os.path.join(td, "s2.fq"), m2[:n8], "r", 2)
            t0 = time.time()
            ob.run_ref(kopt + ["-b160M", "-H4", "s1.fq", "s2.fq"], cwd=td, threads=1)
            wall1 = time.time() - t0
            kmers1 = 2 * n8 * (L - k + 1)
        return {"value": kmers / wall / 1e6, "unit": "Mk-mers/s", "cores": cores, "kind": "reference",
                "sample": sample + "; whole-binary wall %.1f s incl. FASTQ parse and %.1f s fixed start-up" % (wall, startup),
                "value_excl_startup": kmers / max(wall - startup, 1e-9) / 1e6,
                "value_1_thread": kmers1 / max(wall1 - startup, 1e-9) / 1e6,
                "sample_1_thread": "the first eighth of that read set (%d pairs), -j1, start-up excluded" % n8}
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    o = ob.Oracle(k, bloom_bytes=160 << 20)
    t0 = time.time()
    o.load(buf, off)
    o.assemble(buf, off)
    wall = time.time() - t0
    return {"value": kmers / wall / 1e6, "unit": "Mk-mers/s", "cores": 1, "kind": "port", "sample": sample}


def aggregate(elapsed: float, kmers_local: int, steps: int, world: int, device=None):
    """Whole-job numbers from per-rank ones: time = MAX over ranks of the timed region, units =
    SUM over ranks (every rank processes its own read set).  Returns (seconds, total k-mers)."""
    if world == 1:
        return elapsed, kmers_local * steps
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([kmers_local * steps], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def launcher_command(gpus: int, argv, port: str = None):
    """What `python bench.py --gpus N` turns itself into when no launcher set WORLD_SIZE: one rank per GPU over RCCL,
    the way the driver starts an N-GPU run."""
    port = port or os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[N]: 1 = E. coli-scale (5 M pairs, k=64, B=2G; the default and the benchmark), "
                         "2 = human-chr-scale (200 M pairs, k=64, B=40G), 3 = spaced seed (5 M pairs, -k96 -K32, B=2G)")
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per GPU (overrides --config)")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--bloom", type=str, default=None)
    ap.add_argument("--K", type=int, default=None, help="spaced seed of two K-mers (-K of abyss-bloom-dbg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", dest="end_to_end", action="store_false",
                    help="skip the drop-in binary's FASTQ-in / FASTA-out run on this read set (about 20 s: writes the two FASTQ files)")
    ap.add_argument("--invariants", action="store_true",
                    help="after the timed steps, add digests of the last step's device state to the line (popcounts of the counting filter, "
                         "sha256 of the counting and the visited filter, read and unitig counters): what two runs of one workload -- "
                         "e.g. plain and ABG_FORCE_DIST=1 -- must agree on where no reference run exists (configs[2])")
    ap.add_argument("--no-events", action="store_true",
                    help="time the steps without HIP events around every launch (no per-kernel numbers: shows what the events cost)")
    ap.add_argument("--mode", choices=["partitioned", "replicas"], default="partitioned",
                    help="N > 1: one job with the filter partitioned over the ranks (strong scaling) or N independent jobs")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="partitioned mode: the fixed --config / --pairs / --bloom job over N ranks (strong: the default -- "
                         "north_star's experiment is BASELINE.json configs[2] over 1/2/4/8 GPUs), or a job N times as big "
                         "(weak: --pairs and --bloom are per rank)")
    ap.add_argument("--chunks", type=int, default=None,
                    help="feed each pass its reads in this many calls (abg_load_packed / abg_assemble_packed carry their state from call "
                         "to call): bounds the reads a rank holds at once -- in a partitioned run the all-gathered share of a chunk "
                         "instead of the whole read set.  Default 1; 8 for --config 4")
    ap.add_argument("--slice-filter", action="store_true",
                    help="partitioned runs: each rank keeps its own range of the counting filter only (abg_params.slice_filter = 1; "
                         "without the flag the library decides by the device's memory)")
    ap.add_argument("--no-one-gpu-point", dest="one_gpu_point", action="store_false",
                    help="strong-scaled N > 1 runs: do not measure the one-GPU point of the same job in this run (rank 0 alone, one warm-up "
                         "and one timed step before the partitioned steps: about two minutes for configs[2]); the point on file is replayed instead")
    ap.add_argument("--comm", choices=["rccl", "staged"], default="rccl",
                    help="partitioned mode: the library's RCCL communicator, or torch.distributed on host copies (diagnosis)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher
        return subprocess.call(launcher_command(a.gpus, sys.argv[1:]),
                               env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    if a.gpus != int(os.environ.get("WORLD_SIZE", "1")) and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %s ranks: timing what was started\n" % (a.gpus, os.environ.get("WORLD_SIZE", "1")))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if a.config is None:
        # one GPU: configs[1], the configuration the metric is quoted on.  Several GPUs, one partitioned job: configs[2], the
        # configuration BASELINE.json lists for the partitioned filter ("200 M x 2x150 bp, k=64, B=40G ... over 8xMI355X via RCCL"),
        # strong-scaled; its one-GPU point is on file (profiles/r04_c_bench_config2_invariants.json)
        a.config = 2 if (world_env > 1 and a.mode == "partitioned" and a.pairs is None) else 1
    preset = {1: (5_000_000, 64, "2G", 0, "E. coli-scale"), 2: (200_000_000, 64, "40G", 0, "human-chr-scale"),
              3: (5_000_000, 96, "2G", 32, "spaced-seed"),
              # configs[4]: B = 500G is beyond one GPU -- a sliced filter over the ranks (abg_params.slice_filter decides that by
              # itself), the reads in chunks.  Never run at size (no 8-GPU node so far); `--config 4 --pairs .. --bloom ..` scales it down
              4: (1_200_000_000, 96, "500G", 0, "human-scale")}[a.config]
    if a.chunks is None:
        a.chunks = 8 if a.config == 4 else 1
    a.pairs = preset[0] if a.pairs is None else a.pairs
    a.k = preset[1] if a.k is None else a.k
    a.bloom = preset[2] if a.bloom is None else a.bloom
    a.K = preset[3] if a.K is None else a.K
    workload_name = preset[4]

    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")  # RCCL's banner / logs: not on stdout next to the JSON line
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)  # before the process group: RCCL binds the communicator to the current device
    backend = os.environ.get("ABG_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm; ABG_BENCH_BACKEND=gloo lets the launch path be exercised on a
        # box with fewer GPUs than ranks (ranks then share devices)
        import datetime
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=60))  # (the other ranks wait while rank 0 measures the job's one-GPU point)
    device = torch.device("cuda", local)

    mult = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30}
    bloom_bytes = int(float(a.bloom[:-1]) * mult[a.bloom[-1].upper()]) if a.bloom[-1].isalpha() else int(a.bloom)
    read_len, cov, err = 150, 50.0, 0.005
    partitioned = world > 1 and a.mode == "partitioned"
    single = (a.pairs, bloom_bytes, a.bloom)
    if partitioned and a.scaling == "weak":
        # one job, N times the reads, genome and filter of the single-GPU workload
        a.pairs *= world
        bloom_bytes *= world
        a.bloom = "%dx%s" % (world, a.bloom)
    genome_len = int(a.pairs * 2 * read_len / cov)
    comm = None
    comm_note = ""
    if world == 1 and os.environ.get("ABG_FORCE_DIST", "0") not in ("", "0"):
        # the partitioned code path on one rank (every collective an identity): what that path costs by itself
        from abyss_amd import dist as adist
        partitioned = True
        comm = adist.RcclComm(local, single=True)
        comm_note = "ABG_FORCE_DIST: partitioned code path on a single rank"
    elif partitioned:
        from abyss_amd import dist as adist
        if a.comm == "rccl":
            # the library's own RCCL communicator; its id travels through the process group.  If RCCL
            # cannot be brought up on ANY rank, all ranks fall back to independent replicas together.
            try:
                comm = adist.RcclComm(local)
                ok = 1
            except Exception as e:  # noqa: BLE001
                comm_note = "rccl communicator failed (%r): replicas instead" % (e,)
                ok = 0
            t = torch.tensor([ok], dtype=torch.int32, device=device if backend == "nccl" else None)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                if comm is not None:
                    comm.close()
                comm, partitioned = None, False
                comm_note = comm_note or "rccl communicator failed on another rank: replicas instead"
                a.pairs, bloom_bytes, a.bloom = single
                genome_len = int(a.pairs * 2 * read_len / cov)
    # The communicator checked against the host before the job trusts it with its filter (abyss_amd.dist.selftest: uneven and empty
    # all-to-all parts, in-place all-gather, every reduction the engine uses).  A failure on any rank turns the run into replicas.
    selftest = None
    if partitioned and world > 1 and comm is not None:
        t_st = time.perf_counter()
        try:
            selftest = adist.selftest(comm, lambda n: adist.TorchBuf(n, device), stream=None, sync=torch.cuda.synchronize)
        except Exception as e:  # noqa: BLE001
            selftest = {"ok": False, "checks": 0, "failed": ["exception: %r" % (e,)]}
        selftest["seconds"] = round(time.perf_counter() - t_st, 2)
        t = torch.tensor([1 if selftest["ok"] else 0], dtype=torch.int32, device=device if backend == "nccl" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        selftest["all_ranks_ok"] = bool(int(t.item()))
        if not selftest["all_ranks_ok"]:
            sys.stderr.write("bench.py: rank %d: communicator self-test FAILED (%r): replicas instead\n" % (rank, selftest["failed"]))
            comm.close()
            comm, partitioned = None, False
            comm_note = "rccl communicator failed its self-test: replicas instead"
            a.pairs, bloom_bytes, a.bloom = single
            genome_len = int(a.pairs * 2 * read_len / cov)
    # The one-GPU point of THIS job, measured in THIS run: rank 0 alone, the whole read set, the plain (unpartitioned) engine, one
    # warm-up and one timed step; the other ranks wait.  (A driver that divides the N-GPU value by N times ITS one-GPU line --
    # configs[1] by default -- compares two workloads: `scaling_base` below says which job the N-GPU value belongs to.)
    one_gpu = None
    if partitioned and world > 1 and a.scaling == "strong" and a.one_gpu_point and bloom_bytes <= (100 << 30):
        if rank == 0:
            try:
                t_g = time.perf_counter()
                w1, o1, l1 = gen_packed_reads(genome_len, a.pairs, read_len, err, seed=42, device=device)
                g1 = api.BloomDBG(a.k, bloom_bytes=bloom_bytes, num_hashes=4, min_cov=2, device=local,
                                  spaced_seed=api.spaced_seed_kmer_pair(a.k, a.K) if a.K else None)
                nr1 = 2 * a.pairs
                ms1 = []
                for it in range(2):
                    if it:
                        g1.reset()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    g1.load_packed(w1.data_ptr(), o1.data_ptr(), l1.data_ptr(), nr1)
                    g1.assemble_packed(w1.data_ptr(), o1.data_ptr(), l1.data_ptr(), nr1, want_results=False, want_contigs=False)
                    ms1.append((time.perf_counter() - t1) * 1e3)
                c1 = g1.assembly_counters()
                one_gpu = {"value": nr1 * (read_len - a.k + 1) / (ms1[-1] / 1e3) / 1e6, "ms_per_step": ms1[-1], "warmup_ms": ms1[0],
                           "unitigs": c1["next_contig_id"], "unitig_bp": c1["bases_assembled"], "measured": True,
                           "what": "rank 0 alone on the whole job (plain engine), one warm-up + one timed step, in this run",
                           "seconds_spent": round(time.perf_counter() - t_g, 1)}
                g1.close()
                del w1, o1, l1, g1
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001  (out of memory for this job on one GPU, ...: the point on file is replayed)
                sys.stderr.write("bench.py: one-GPU point not measured: %r\n" % (e,))
                one_gpu = None
                torch.cuda.empty_cache()
        torch.cuda.synchronize()
        dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
    if partitioned:
        # one job: rank r holds the r-th block of the read set (same genome on every rank)
        pairs_local = a.pairs // world + (1 if rank < a.pairs % world else 0)
        first_pair = rank * (a.pairs // world) + min(rank, a.pairs % world)
        words, woff, lens = gen_packed_reads(genome_len, pairs_local, read_len, err, seed=42, device=device,
                                             first=first_pair, total_pairs=a.pairs)
        n_reads = 2 * pairs_local
    else:
        # (replicas: every rank its own read set; rank 0's is the pinned one)
        words, woff, lens = gen_packed_reads(genome_len, a.pairs, read_len, err, seed=42, device=device, read_seed=7 + rank)
        n_reads = 2 * a.pairs
    golden = golden_for(a.config, a.pairs, a.k, a.K, a.bloom) if (world == 1 or (partitioned and a.scaling == "strong")) else None
    kmers = n_reads * (read_len - a.k + 1)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
            torch.cuda.synchronize()

    g = None
    unitigs = bases = 0
    setup_s = 0.0
    phase_s = [0.0, 0.0]  # wall seconds in PASS 1 / PASS 2 over the timed steps

    def step(profile=True):
        nonlocal g, unitigs, bases, setup_s, comm
        t_setup = time.perf_counter()
        if g is None:
            g = api.BloomDBG(a.k, bloom_bytes=bloom_bytes, num_hashes=4, min_cov=2, device=local,
                             spaced_seed=api.spaced_seed_kmer_pair(a.k, a.K) if a.K else None,
                             **({"slice_filter": 1} if a.slice_filter and partitioned else {}))
            if partitioned:
                if comm is None:  # --comm staged
                    comm = adist.StagedTorchComm(*adist.device_memory_io(g), group=dist.new_group(backend="gloo"))
                g.attach_comm(comm)
        else:
            g.reset()  # empty filters, zero counters; the device memory is kept (abg_reset)
        setup_s += time.perf_counter() - t_setup
        g.profile_enable(profile)
        if profile:
            g.profile_reset()
        def reads_of(c):
            # chunk c of this rank's reads (the words stay where they are: offsets index them) -- in a partitioned run
            # all-gathered, part of the step (every rank walks and guides over every read of the chunk)
            lo, hi = n_reads * c // a.chunks, n_reads * (c + 1) // a.chunks
            wp, op, lp = words.data_ptr(), woff.data_ptr() + 8 * lo, lens.data_ptr() + 4 * lo
            return g.share_reads(wp, op, lp, hi - lo) if partitioned else (wp, op, lp, hi - lo)
        t_a = time.perf_counter()
        for c in range(a.chunks):
            rw, ro, rl, rn = reads_of(c)
            g.load_packed(rw, ro, rl, rn)
        t_b = time.perf_counter()
        # contigs stay on the device (no per-contig callback into Python): the unitig count and
        # their total length come from the assembly counters (AssemblyCounters.h:15-31)
        for c in range(a.chunks):
            if a.chunks > 1 or not partitioned:
                rw, ro, rl, rn = reads_of(c)  # (one chunk: what PASS 1 gathered is still there)
            g.assemble_packed(rw, ro, rl, rn, want_results=False, want_contigs=False)
        phase_s[0] += t_b - t_a  # (both calls return with the device idle)
        phase_s[1] += time.perf_counter() - t_b
        c = g.assembly_counters()
        unitigs, bases = c["next_contig_id"], c["bases_assembled"]

    # N = 1: every launch of the timed steps is bracketed by HIP events (the roofline numbers are
    # measured live).  Partitioned: the events would put a host synchronisation behind every
    # launch and collective, so the per-kernel split comes from the warm-up steps instead.
    timed_profile = not partitioned
    for _ in range(a.warmup):
        step()
    warm_prof = None
    if partitioned and a.warmup and g is not None:
        warm_prof = True
    names = ["hash_bin_staged", "hash_ops", "bin_coarse", "bin_fine", "tile_purity", "op_target", "dist_pack", "tile_apply", "claim_list", "guide_build",
             "hash_claim", "insert_round", "insert_retry", "insert_apply", "compact", "drain_vals", "drain_load",
             "insert_drain", "classify", "read_prep", "presearch_scan", "presearch", "walk", "rewalk", "merge_fix", "comm_all_reduce", "comm_all_gather",
             "share_fix", "route_pack", "route_reply", "route_tgt", "route_pend", "comm_all_to_all", "dup_link",
             "reclassify", "contig_prep", "predict", "precommit", "commit", "pc_count", "pc_stamp", "pc_short", "pc_timemin",
             "pc_decide", "pc_break", "pc_apply", "pc_write", "popcount"]
    prof = {nm: g.profile_get(nm) for nm in names} if (warm_prof and g is not None) else None
    barrier()
    phase_s[0] = phase_s[1] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(profile=timed_profile and not a.no_events)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event timings of the last timed step (events are recorded on the library's stream)
    if prof is None:
        prof = {nm: g.profile_get(nm) for nm in names}
    red_dev = device if (world > 1 and dist.get_backend() == "nccl") else None
    elapsed, total_kmers = aggregate(elapsed, kmers, a.steps, world, red_dev)
    timed_pass_s = list(phase_s)
    # Parity of the timed workload: its unitig count and total length against the reference's -j1 run on
    # this very read set (tests/golden/full_size.json; the FASTA's bytes are compared by
    # tests/test_gpu_fullsize.py and by the end-to-end leg below).  A fast wrong answer is no result.
    parity = None
    if golden is not None:
        parity = {"reference": "oracle/_ref/abyss-bloom-dbg -j1 on the FASTQ files of this read set (tests/golden/full_size.json)",
                  "unitigs": [unitigs, golden["unitigs"]], "unitig_bp": [bases, golden["unitig_bp"]],
                  "ok": bool(unitigs == golden["unitigs"] and bases == golden["unitig_bp"])}
        if not parity["ok"] and rank == 0:
            sys.stderr.write("bench.py: PARITY FAILURE: %r\n" % (parity,))
    # the same steps without a HIP event pair (and its host synchronisation) around every launch: what the
    # library achieves when nobody is measuring its kernels
    no_events = None
    if timed_profile and not a.no_events and world == 1:
        k2 = max(1, min(a.steps, 3))
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            step(profile=False)
        barrier()
        e2 = time.perf_counter() - t1
        no_events = {"value": kmers * k2 / e2 / 1e6, "ms_per_step": e2 / k2 * 1e3, "steps": k2}
    phase_s[0], phase_s[1] = timed_pass_s

    stats = g.stats()
    ranks_agree = None
    if partitioned and world > 1:
        box = [None] * world
        dist.all_gather_object(box, (unitigs, bases, stats["insert_rounds"], stats["candidates"]))
        ranks_agree = all(b == box[0] for b in box)
    kmers_all = kmers  # k-mer ops whose pairs a rank scans per step (all of the job's when partitioned: every rank picks its own counters' pairs out of all ops)
    if partitioned:
        kmers_all = 2 * a.pairs * (read_len - a.k + 1)
    H = 4
    per_kmer_bases = (read_len / 4.0) / (read_len - a.k + 1)
    unitig_kmers = max(bases - unitigs * (a.k - 1), 0)
    # Algorithmic bytes of one step per kernel family (SURVEY.md 8d terms):
    # per read k-mer for the streaming kernels, per unitig k-mer for the walk and the commit.
    # Partitioned: a rank hashes every op of the job but touches only its 1/world of the claim slots
    # and counters, classifies 1/world of the reads and walks 1/world of the candidates.
    share = 1.0 / world if partitioned else 1.0
    # SURVEY.md 8d, per read k-mer: 2-bit bases (one pass), 2H for the insert (H counter reads + H writes), H + H for
    # the solid / visited membership tests (+ the blunt-end look-ahead); per unitig k-mer: 8 neighbour queries x H
    # for the walk, 3 commit passes x H + H writes.  The kernels of one family share its bytes: a family's
    # achieved rate = its algorithmic bytes / the summed duration of its kernels.
    families = {
        "pass1": (["hash_bin_staged", "hash_ops", "bin_coarse", "bin_fine", "tile_purity", "op_target", "dist_pack", "tile_apply", "claim_list", "hash_claim",
                   "insert_round", "insert_retry", "insert_apply", "compact", "drain_vals", "drain_load", "insert_drain", "route_pack", "route_reply", "route_tgt", "route_pend"],
                  (per_kmer_bases + 2 * H * share) * kmers_all),
        "classify": (["classify", "reclassify"],
                     (per_kmer_bases + 2 * H + 2 * 5 * 4 * H / (read_len - a.k + 1)) * kmers_all * share),
        "walk": (["presearch_scan", "presearch", "walk", "rewalk"], 8 * H * unitig_kmers * share),
        "commit": (["contig_prep", "dup_link", "precommit", "commit", "pc_count", "pc_stamp", "pc_short", "pc_timemin", "pc_decide",
                    "pc_break", "pc_apply", "pc_write"], 4 * H * unitig_kmers),
    }
    per_kernel = {}
    for fam, (members, total) in families.items():
        ms = sum(prof[nm][0] for nm in members)
        n = max([prof[nm][1] for nm in members] + [0])
        if ms > 0:
            if fam == "pass1" and prof["hash_bin_staged"][1]:
                # the next batch's hashing and binning run on the side stream BESIDE this batch's reservation
                # rounds ("hash_bin_staged" = that stretch's elapsed time): the overlapped time is counted once,
                # i.e. the family's duration is the pass's wall time
                ms = min(ms, phase_s[0] / a.steps * 1e3)
            gbs = total / 1e9 / (ms / 1e3)
            top = max([nm for nm in members if nm != "hash_bin_staged"], key=lambda nm: prof[nm][0])
            per_kernel[fam] = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "ms": ms, "algorithmic_GB": total / 1e9,
                               "longest_kernel": top, "avg_launch_ms": prof[top][0] / max(prof[top][1], 1),
                               "launches": prof[top][1]}
    if not per_kernel:
        # (partitioned run without a warm-up step: no per-launch events were taken -- see timed_profile)
        per_kernel = {"walk": {"achieved": None, "frac": None, "avg_launch_ms": None, "launches": 0, "ms": 0, "longest_kernel": "rewalk"}}
    # the dominant kernel: the one with the largest summed duration over the step among the kernels of the MAIN stream; the roofline
    # line is its family's.  (The classification runs on the side stream beside the walkers, at the lowest priority: the time between
    # its events is not its own -- 87 ms of kernel when it runs alone, 165-185 beside them, within a few ms of the walkers' own sum,
    # so that which of the two was "longest" changed from box to box.  It stays in `kernels`.)
    main_stream = [fam for fam in per_kernel if fam != "classify"] or list(per_kernel)
    dom = max(main_stream, key=lambda fam: prof[per_kernel[fam]["longest_kernel"]][0] if per_kernel[fam].get("longest_kernel") in prof else 0)
    # HBM bytes from the TCC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this very
    # command, tools/gpu_pmc_traffic.sh; KB units, uncalibrated for narrow random accesses: MI355X_MICROARCH.md),
    # committed with the commit they were taken at; per launch of the family's longest kernel like `achieved`
    traffic = traffic_src = None
    traffic_head = traffic_behind = None
    tsrc, tj, traffic_head, traffic_behind = pick_evidence("_pmc_traffic.json")
    if tj is not None and a.config == 1 and a.pairs == 5_000_000 and world == 1:
        kname = {"rewalk": "FWalk", "walk": "FWalk", "tile_apply": "FTileApply", "tile_purity": "FTilePurity",
                 "bin_coarse": "FBinCoarse", "bin_fine": "FBinFine", "op_target": "FOpTarget", "hash_ops": "FHashOps",
                 "insert_round": "FInsertRound", "insert_retry": "FInsertRound", "hash_claim": "FHashClaim",
                 "classify": "FClassify", "pc_timemin": "FPcTimeMin", "pc_decide": "FPcDecide"}.get(per_kernel[dom]["longest_kernel"])
        t = tj.get(kname)
        if t:
            traffic = (t["FETCH_SIZE"]["sum"] + t["WRITE_SIZE"]["sum"]) * 1024 / max(t["FETCH_SIZE"]["dispatches"], 1)
            traffic_src = "%s (%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this command, KB units, per launch; taken at commit %s)" % (tsrc, kname, tj.get("_commit", "?"))
    step_bytes = (2 * per_kmer_bases + 4 * H + 2 * 5 * 4 * H / (read_len - a.k + 1)) * kmers + 12 * H * unitig_kmers
    roofline = {"bound": "hbm", "kernel": per_kernel[dom]["longest_kernel"], "family": dom,
                "achieved": per_kernel[dom]["achieved"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": traffic, "traffic_source": traffic_src,
                # (taken on exactly these kernel sources?  else: commits to abyss_amd/csrc since -- None where git is not there to ask)
                "traffic_kernels_are_head": traffic_head, "traffic_csrc_commits_behind": traffic_behind,
                # (the rocprofv3 --kernel-trace --stats summary taken in the same gpurun call as the traffic file: tools/gpu_r5_final.sh counters)
                "kernel_stats": (lambda f: f if tsrc and os.path.exists(os.path.join(ROOT, f)) else None)((tsrc or "").replace("_pmc_traffic.json", "_kernel_stats_config1.csv")),
                "avg_launch_ms": per_kernel[dom]["avg_launch_ms"], "launches": per_kernel[dom]["launches"],
                "note": "the main stream's kernel with the largest summed duration of the step (the classification overlaps the walk on a side stream: its family is in `kernels`); "
                        "achieved = algorithmic bytes of its family (SURVEY.md 8d) over the summed duration of the family's kernels; "
                        "the walk is a graph traversal of dependent random probes, not a stream (DESIGN.md section 4.2)",
                "whole_step": {"algorithmic_GB": step_bytes / 1e9, "achieved": step_bytes / 1e9 / (elapsed / a.steps),
                               "frac": step_bytes / 1e9 / (elapsed / a.steps) / HBM_PEAK_GBS},
                "kernels": per_kernel}

    if rank == 0:
        out = {
            "metric": METRIC, "value": total_kmers / elapsed / 1e6, "unit": "Mk-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            # one GPU: nothing is scaled; N ranks of one partitioned job: as asked; N independent replicas: weak by construction
            "scaling": None if world == 1 else (a.scaling if partitioned else "weak"),
            "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s synthetic: %d x 2x%d bp reads, k=%d%s, B=%s, H=4, %s"
                       % (workload_name, a.pairs, read_len, a.k, (" K=%d spaced seed" % a.K) if a.K else "", a.bloom,
                          "one job over %d MI355X" % world if partitioned else "1xMI355X per rank"),
                       "baseline_config": a.config, "genome_bp": genome_len, "coverage": cov, "error_rate": err,
                       "read_kmers": kmers_all if partitioned else kmers,
                       "parallelism": ("filter range-partitioned over %d ranks in PASS 1 (%s), gathered for PASS 2; walks split"
                                       % (world, ("RCCL on the engine's stream: (op, counter) pairs routed to the owning ranks by all-to-all (ncclSend/ncclRecv groups)" if world >= 4 else "RCCL all_gather + all_reduce on the engine's stream") if a.comm == "rccl"
                                          else "torch.distributed on host copies")) if partitioned
                       else ("replicas x%d (independent jobs)" % world if world > 1 else "single GPU"),
                       "unitigs": unitigs, "unitig_bp": bases, **({"chunks": a.chunks} if a.chunks > 1 else {})},
            "roofline": roofline,
            "kernel_ms": {nm: {"ms": round(v[0], 3), "launches": v[1]} for nm, v in prof.items() if v[1]},
            "engine_stats": stats,
            # part of every step: creating the filters (first step) or clearing them (abg_reset)
            "setup_ms_per_step": round(setup_s / max(a.steps + a.warmup, 1) * 1e3, 1),
            "pass_ms_per_step": {"pass1": round(phase_s[0] / a.steps * 1e3, 1), "pass2": round(phase_s[1] / a.steps * 1e3, 1)},
        }
        if parity is not None:
            out["parity"] = parity
        if no_events is not None:
            out["no_events"] = no_events
        if selftest is not None:
            out["rccl_selftest"] = selftest
        if partitioned and world > 1 and a.scaling == "strong":
            # the one-GPU point of this very job: measured in this run by rank 0 alone (see above), else REPLAYED from a committed
            # earlier run (the driver computes its own efficiency from its per-N lines; this is for a reader of a single line)
            one = {1: "r04_e_bench_default.json", 2: "r04_c_bench_config2_invariants.json", 3: "r04_b_bench_config3_spaced_seed_k96_K32.json"}.get(a.config)
            src = os.path.join(ROOT, "profiles", one) if one else None
            if one_gpu is not None:
                v1 = one_gpu["value"]
                out["strong_scaling"] = dict(one_gpu, one_gpu_value=v1, replayed=False, speedup=out["value"] / v1, efficiency=out["value"] / v1 / world,
                                             same_unitigs=bool(one_gpu["unitigs"] == unitigs and one_gpu["unitig_bp"] == bases))
            elif src and os.path.exists(src) and a.pairs == preset[0] and a.k == preset[1]:
                v1 = json.load(open(src))["value"]
                out["strong_scaling"] = {"one_gpu_value": v1, "one_gpu_source": "profiles/" + one, "replayed": True, "measured": False,
                                         "speedup": out["value"] / v1, "efficiency": out["value"] / v1 / world}
            # which job this line's value belongs to: NOT the one a default `--gpus 1` line times (configs[1]) unless --config says so
            out["scaling_base"] = {"config": a.config, "workload": out["config"]["workload"],
                                   "one_gpu_value": out.get("strong_scaling", {}).get("one_gpu_value"),
                                   "one_gpu_measured_in_this_run": one_gpu is not None,
                                   "note": "strong scaling of ONE job over the ranks; `python bench.py --gpus 1` without --config times configs[1], a "
                                           "different job -- divide this value by N x scaling_base.one_gpu_value, not by N x that line"}
        if partitioned:
            out["config"]["ranks_agree"] = ranks_agree
            out["config"]["counter_bytes_per_rank"] = g.stats()["counter_bytes_held"]
            out["kernel_ms_note"] = "rank 0, last warm-up step (the timed steps run without per-launch events)"
        if comm_note:
            out["config"]["note"] = comm_note
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.k, cores=os.cpu_count() or 1, K=a.K)
            # like for like on the FULL read set: the unmodified reference at -j<all cores> is a 4-6 minute run, measured once on a
            # GPU box's host ON THE READ SET TIMED HERE (tools/gpu_cpu_reference_full.sh) and REPLAYED from the committed file
            # (two such runs are on file and they differ by box: 233 s in round 2, 362 s in round 4 -- ratios are to be quoted against
            # the in-run sample above, or against both)
            if a.config == 1 and a.pairs == 5_000_000:
                full = []
                for f in ("r02_cpu_reference_config1.json", "r04_cpu_reference_config1.json"):
                    src = os.path.join(ROOT, "profiles", f)
                    if os.path.exists(src):
                        full.append(dict(json.load(open(src)), replayed=True, source="profiles/" + f))
                if full:
                    out["cpu_baseline"]["reference_full_config"] = full[-1]
                    out["cpu_baseline"]["reference_full_config_all_runs"] = full
        if a.invariants and g is not None:
            import hashlib
            pc, fpc = g.counting_stats()
            vis = g.visited()
            c = g.assembly_counters()
            out["invariants"] = {"popcount": pc, "filtered_popcount": fpc,
                                 # (the counter array itself only while it is a few GB: 38 GB of host memory for a digest is not worth a dead box)
                                 "counters_sha256": hashlib.sha256(g.counters()).hexdigest() if g.size <= (4 << 30) else None,
                                 "visited_sha256": hashlib.sha256(vis).hexdigest(),
                                 "assembly_counters": {k: int(v) for k, v in c.items()}}
            del vis
        if world == 1 and a.end_to_end and a.config in (1, 3):
            # (the binaries run as processes of their own: this process lets go of its ~27 GB of the device first -- they are timed as a
            # user would run them, not beside another tenant of the GPU)
            if g is not None:
                g.close()
                g = None
            words = woff = lens = None
            torch.cuda.empty_cache()
            out["end_to_end"] = end_to_end(a, genome_len, read_len, err, device, golden)
        # whatever native libraries buffered on stdout (RCCL's version banner) goes out first: the
        # JSON line stays a line of its own
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if g is not None:
        g.close()
    if comm is not None and hasattr(comm, "close"):
        comm.close()
    if world > 1:
        dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()
    return 0 if (parity is None or parity["ok"]) else 1


if __name__ == "__main__":
    sys.exit(main())
