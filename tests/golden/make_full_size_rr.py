#!/usr/bin/env python3
"""The stage after AdjList at BASELINE.json's configs[1] size: digests of the UNMODIFIED reference's abyss-rresolver-short
(oracle/_ref, -j1: RResolver/*.cpp over oracle/shim/btllib -- parity with a real btllib build is unpinned, see that directory) on the
full-size read set, its unitigs and their overlap graph.  The unitigs of the reference at -j1 are 21 minutes of CPU away; their
sha256 is pinned in tests/golden/full_size.json and the drop-in abyss-bloom-dbg reproduces it in a second on a GPU, so this script
runs ON A GPU BOX (gpurun), where oracle/_ref travels prebuilt:

    gpurun -- 'python tests/golden/make_full_size_rr.py gpurun_out/r6b'

It refuses to go on unless the unitigs and the graph have the pinned digests; the result (gpurun_out/.../full_size_rr.json) is merged
by hand into the matching run of tests/golden/full_size.json under "rresolver"."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abyss_amd import build, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "abyss-rresolver-short")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    out_dir = sys.argv[1]
    os.makedirs(out_dir, exist_ok=True)
    golden = [g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))["runs"]
              if g.get("generator") == "make_read_set_cb" and g["config"] == 1 and g["bloom"] == "2G"][0]
    k, bloom, pairs = golden["k"], golden["bloom"], golden["pairs"]
    h1, h2 = synth.make_genome(30_000_000, seed=42)
    ncpu = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        m1, m2 = synth.sample_pairs_cb(h1, h2, pairs, read_len=150, err=0.005, seed=7)
        synth.write_fastq(os.path.join(td, "r1.fq"), m1, "r", 1)
        synth.write_fastq(os.path.join(td, "r2.fq"), m2, "r", 2)
        del m1, m2
        cli = build.build_cli()
        fa = subprocess.run([cli, "-k%d" % k, "-b" + bloom, "-H4", "-q3", "-j%d" % ncpu, "r1.fq", "r2.fq"], cwd=td, stdout=subprocess.PIPE, check=True).stdout
        open(os.path.join(td, "asm-1.fa"), "wb").write(fa)
        assert hashlib.sha256(fa).hexdigest() == golden["fasta_sha256"], "the unitigs are not the reference's"
        dot = subprocess.run([os.path.join(build.BIN_DIR, "AdjList")] + golden["adjlist"]["options"].split() + ["asm-1.fa"], cwd=td, stdout=subprocess.PIPE, check=True).stdout
        open(os.path.join(td, "asm-1.dot"), "wb").write(dot)
        assert hashlib.sha256(dot).hexdigest() == golden["adjlist"]["dot_sha256"], "the overlap graph is not the reference's"
        res = {"options": "-b%s -f0.8 -k%d -h asm-1-rr --dot -c asm-1-rr.fa -g asm-1-rr.dot asm-1.fa asm-1.dot r1.fq r2.fq" % (bloom, k)}

        def run(exe, tag, threads):
            for f in os.listdir(td):
                if f.startswith("asm-1-rr"):
                    os.remove(os.path.join(td, f))
            t0 = time.time()
            r = subprocess.run([exe, "-b" + bloom, "-f0.8", "-j%d" % threads, "-k%d" % k, "-h", "asm-1-rr", "--dot", "-c", "asm-1-rr.fa", "-g", "asm-1-rr.dot",
                                "asm-1.fa", "asm-1.dot", "r1.fq", "r2.fq"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            files = {f: sha(os.path.join(td, f)) for f in sorted(os.listdir(td)) if f.startswith("asm-1-rr")}
            res[tag] = {"rc": r.returncode, "wall_s": round(wall, 2), "threads": threads, "sha256": files,
                        "contigs": open(os.path.join(td, "asm-1-rr.fa"), "rb").read().count(b">") if r.returncode == 0 else None}
            if r.returncode != 0:
                res[tag]["stderr"] = r.stderr.decode()[-500:]
            print(tag, res[tag]["rc"], res[tag]["wall_s"], "s", flush=True)
        run(REF, "reference_j1", 1)
        run(REF, "reference_jN", ncpu)
        run(os.path.join(build.BIN_DIR, "abyss-rresolver-short"), "drop_in", ncpu)
        run(os.path.join(build.BIN_DIR, "abyss-rresolver-short"), "drop_in_again", ncpu)
        res["unitigs"] = fa.count(b">")
        res["drop_in_matches_reference_j1"] = res["drop_in"]["sha256"] == res["reference_j1"]["sha256"]
        res["cpu"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    json.dump(res, open(os.path.join(out_dir, "full_size_rr.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
