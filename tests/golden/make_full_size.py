#!/usr/bin/env python3
"""Digests of the UNMODIFIED reference (oracle/_ref/abyss-bloom-dbg, `make -C oracle ref`) run at -j1
-- the deterministic order the GPU path reproduces -- on BASELINE.json's full-size read sets, written
to tests/golden/full_size.json.  The reference takes 20-25 minutes per configuration on one core, so
this is run by hand, in the build container (it needs /root/reference through oracle/_ref):

    python tests/golden/make_full_size.py /tmp/full2        # generates the reads, runs the reference
    python tests/golden/make_full_size.py /tmp/full2 --only ref_c2regime   # one run; its digest is merged into the file

The read set is synth.make_read_set_cb(30_000_000, 50): the counter-based generator whose torch twin
(synth.packed_reads_torch) produces the same reads on the GPU for bench.py and
tests/test_gpu_fullsize.py, so nothing but these digests has to travel.
"""
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abyss_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "abyss-bloom-dbg")
RUNS = [  # (tag, BASELINE.json config, k, K, bloom)
    ("ref_c1", 1, 64, 0, "2G"),
    ("ref_c3", 3, 96, 32, "2G"),
    # configs[2]'s own regime at configs[1]'s size: the same 5 M pairs with the 35.8 filter bytes per genome base that B=40G gives a
    # 1.2 Gbp genome (1 GiB / 30 Mbp) -- filtered occupancy ~18 %, "Bloom filter FPR" ~20 % per hash function.  configs[2] itself
    # (200 M pairs) is 14-15 hours of the reference at -j1 and 46 GB of memory: beyond a round of this build
    ("ref_c2regime", 2, 64, 0, "1G"),
]
GENOME, COV, PAIRS = 30_000_000, 50.0, 5_000_000


def digest(tag, wd):
    fa = open(os.path.join(wd, tag + ".fa"), "rb").read()
    err = open(os.path.join(wd, tag + ".err"), "rb").read().decode(errors="replace")
    lens = [int(line.split()[1]) for line in fa.split(b"\n") if line.startswith(b">")]
    info = {"fasta_sha256": hashlib.sha256(fa).hexdigest(), "fasta_md5": hashlib.md5(fa).hexdigest(), "fasta_bytes": len(fa),
            "unitigs": len(lens), "unitig_bp": sum(lens)}
    for line in err.splitlines():
        if "popcount" in line and "=" in line:
            info["filtered_popcount"] = int(line.split("=")[1])  # (counters >= the threshold: printed after PASS 1)
        if line.startswith("Processed") and "solid reads" in line:
            info["last_progress_line"] = line.strip()
    done = os.path.join(wd, tag + ".done")
    if os.path.exists(done):
        info["reference_run"] = open(done).read().strip()
    return info


def main():
    wd = sys.argv[1]
    only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--only" else None
    runs = [r for r in RUNS if only is None or r[0] == only]
    os.makedirs(wd, exist_ok=True)
    if not os.path.exists(os.path.join(wd, "r2.fq")):
        m1, m2 = synth.make_read_set_cb(GENOME, COV)
        synth.write_fastq(os.path.join(wd, "r1.fq"), m1, "r", 1)
        synth.write_fastq(os.path.join(wd, "r2.fq"), m2, "r", 2)

    def run(tag, k, K, bloom):
        if os.path.exists(os.path.join(wd, tag + ".done")):
            return
        t = time.time()
        args = [REF, "-k%d" % k] + (["-K%d" % K] if K else []) + ["-b" + bloom, "-H4", "-q3", "-j1", "-v", "r1.fq", "r2.fq"]
        with open(os.path.join(wd, tag + ".fa"), "wb") as o, open(os.path.join(wd, tag + ".err"), "wb") as e:
            rc = subprocess.call(args, stdout=o, stderr=e, cwd=wd, env=dict(os.environ, OMP_NUM_THREADS="1"))
        open(os.path.join(wd, tag + ".done"), "w").write("rc=%d wall=%ds\n" % (rc, time.time() - t))
    ts = [threading.Thread(target=run, args=(tag, k, K, bloom)) for tag, _, k, K, bloom in runs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    out = {"what": "reference abyss-bloom-dbg 2.3.10 (oracle/_ref, unmodified sources) at -j1 on the full-size synthetic read sets of BASELINE.json; "
                   "made by tests/golden/make_full_size.py",
           "read_set": {"generator": "synth.make_read_set_cb", "genome_bp": GENOME, "coverage": COV, "pairs": PAIRS, "read_len": 150, "error_rate": 0.005,
                        "genome_seed": 42, "read_seed": 7, "ids": "r<i>/1 for the mate-1 file, r<i>/2 for the mate-2 file (synth.write_fastq)"},
           "runs": []}
    path = os.path.join(ROOT, "tests", "golden", "full_size.json")
    if only is not None and os.path.exists(path):
        out = json.load(open(path))  # (one run made anew: the others stay as they are)
    for tag, config, k, K, bloom in runs:
        info = digest(tag, wd)
        info.update({"generator": "make_read_set_cb", "config": config, "pairs": PAIRS, "k": k, "K": K, "bloom": bloom,
                     "options": "-k%d%s -b%s -H4 -q3 -j1" % (k, " -K%d" % K if K else "", bloom)})
        if tag == "ref_c2regime":
            info["note"] = "configs[2]'s filter bytes per genome base (35.8) on the configs[1] read set: the regime of B=40G / 1.2 Gbp at a size the reference finishes"
        same = [i for i, r in enumerate(out["runs"]) if r.get("generator") == "make_read_set_cb" and (r["k"], r.get("K", 0), r["bloom"]) == (k, K, bloom)]
        if same:
            info = dict(out["runs"][same[0]], **info) if only is not None else info
            out["runs"][same[0]] = info
        else:
            out["runs"].append(info)
    # the earlier pin of configs[1]: the sequential generator's read set (synth.make_read_set), whose reference FASTA has the
    # md5 the drop-in binary produced end to end in round 2 (profiles/r02_end_to_end.json)
    legacy = os.path.join(os.path.dirname(wd.rstrip("/")), "full1", "ref_c1.fa")
    if only is None and os.path.exists(legacy):
        fa = open(legacy, "rb").read()
        out["runs"].append({"generator": "make_read_set", "config": 1, "pairs": PAIRS, "k": 64, "K": 0, "bloom": "2G", "options": "-k64 -b2G -H4 -q3 -j1",
                            "fasta_sha256": hashlib.sha256(fa).hexdigest(), "fasta_md5": hashlib.md5(fa).hexdigest(), "fasta_bytes": len(fa),
                            "unitigs": fa.count(b">"), "note": "same md5 as fasta_md5 of profiles/r02_end_to_end.json (the drop-in binary, round 2)"})
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
