#!/usr/bin/env python3
"""Digests of the UNMODIFIED reference (oracle/_ref/abyss-bloom-dbg) at -j1 on a down-scaled replica of
BASELINE.json's configs[2] *in its own regime*: configs[2] gives the counting filter 40 GiB for a
1.2 Gbp genome = 35.8 filter bytes per genome base (filtered occupancy ~18 %), half of configs[1]'s
71.6 B per base, so the false-positive branches, the crowded counters of PASS 1 and the redundancy
tests of the commit all run at a different rate than in any configs[1]-shaped test (SURVEY.md 8d:
"a down-scaled replica with identical parameters-per-genome-base").  Replica: the 400 k pairs / 2.4 Mbp
read set of tests/test_gpu_scale.py (synth.make_read_set, numpy, reproducible on the GPU box), k=64,
H=4, -b82M (82 MiB / 2.4 Mbp = 35.8 B per base).  Written to tests/golden/config2_regime.json;
tests/test_gpu_scale.py drives the drop-in binary through the same run and compares.

    python tests/golden/make_config2_regime.py /tmp/c2regime      # ~5 min on one core

Needs /root/reference through oracle/_ref (`make -C oracle ref`), so it runs in the build container.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abyss_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "abyss-bloom-dbg")
GENOME, COV, PAIRS, BLOOM = 2_400_000, 50.0, 400_000, "82M"
OPTS = ["-k64", "-b" + BLOOM, "-H4", "--checkpoint=400000", "--keep-checkpoint", "--read-log=rl.tsv", "-T", "tr.tsv",
        "--checkpoint-prefix=ck", "r1.fq", "r2.fq"]


def strip_length_column(trace: bytes) -> bytes:
    # (the trace's second column is the path length in vertices *including* the start; test_gpu_scale.py strips it the same way)
    return b"".join(b"\t".join(r.split(b"\t")[:1] + r.split(b"\t")[2:]) + b"\n" for r in trace.splitlines())


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def main():
    wd = sys.argv[1]
    os.makedirs(wd, exist_ok=True)
    m1, m2 = synth.make_read_set(GENOME, COV)
    assert m1.shape[0] == PAIRS
    synth.write_fastq(os.path.join(wd, "r1.fq"), m1, "r", 1)
    synth.write_fastq(os.path.join(wd, "r2.fq"), m2, "r", 2)
    t = time.time()
    with open(os.path.join(wd, "ref.fa"), "wb") as o, open(os.path.join(wd, "ref.err"), "wb") as e:
        rc = subprocess.call([REF, "-j1", "-v"] + OPTS, stdout=o, stderr=e, cwd=wd, env=dict(os.environ, OMP_NUM_THREADS="1"))
    wall = time.time() - t
    assert rc == 0
    fa = open(os.path.join(wd, "ref.fa"), "rb").read()
    err = open(os.path.join(wd, "ref.err"), "rb").read().decode(errors="replace")
    lens = [int(line.split()[1]) for line in fa.split(b"\n") if line.startswith(b">")]
    out = {"what": "reference abyss-bloom-dbg 2.3.10 (oracle/_ref, unmodified sources) at -j1 on a replica of configs[2] at its own "
                   "filter bytes per genome base; made by tests/golden/make_config2_regime.py",
           "read_set": {"generator": "synth.make_read_set", "genome_bp": GENOME, "coverage": COV, "pairs": PAIRS},
           "options": " ".join(["-j1"] + OPTS), "bloom": BLOOM, "filter_bytes_per_genome_base": 82 * 2**20 / GENOME,
           "reference_run": "rc=%d wall=%ds" % (rc, wall),
           "fasta_sha256": sha(fa), "fasta_bytes": len(fa), "unitigs": len(lens), "unitig_bp": sum(lens),
           "readlog_sha256": sha(open(os.path.join(wd, "rl.tsv"), "rb").read()),
           "trace_nolen_sha256": sha(strip_length_column(open(os.path.join(wd, "tr.tsv"), "rb").read())),
           "checkpoint_sha256": {ext: sha(open(os.path.join(wd, "ck" + ext), "rb").read())
                                 for ext in (".dbg.bloom", ".visited.bloom", ".counters.tsv", ".contigs.fa")}}
    for line in err.splitlines():
        if "popcount" in line and "=" in line:
            out["filtered_popcount"] = int(line.split("=")[1])
        if "FPR" in line:
            out.setdefault("fpr_lines", []).append(line.strip())
        if line.startswith("Processed") and "solid reads" in line:
            out["last_progress_line"] = line.strip()
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config2_regime.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
