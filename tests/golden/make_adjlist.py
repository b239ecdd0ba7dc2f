"""Regenerates tests/golden/adjlist/*: outputs of the UNMODIFIED reference AdjList
(oracle/_ref/AdjList, built from /root/reference by oracle/Makefile) on the committed unitig
FASTAs of tests/golden/ and on the synthetic contig sets of tests/test_adjlist.py.

Run in the build container (needs /root/reference): python tests/golden/make_adjlist.py
With `--full-size UNITIGS.fa` it instead runs the reference on a full-size unitig FASTA written by
the reference abyss-bloom-dbg (tests/golden/make_full_size.py keeps it in its work directory),
checks that it is the FASTA tests/golden/full_size.json pins, and records the sha256 of the
`--dot` output there (what abyss-pe asks for: AdjList -k$k -m$m --dot, bin/abyss-pe:238-246,575-577).
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref", "AdjList")

# (case, FASTA, k, -m as abyss-pe passes it (bin/abyss-pe:238-246), extra)
CASES = [
    ("k25_m0", "k25_h3_kc3_t40.fa", 25, 0, []),
    ("k32_m0", "k32.fa", 32, 0, []),
    ("k32_m5_ss", "k32.fa", 32, 5, ["--SS"]),
    ("k40_m0", "k40_mixed.fa", 40, 0, []),
    ("k40_m12", "k40_mixed.fa", 40, 12, []),
    ("k64_m50", "k64.fa", 64, 50, []),
    ("k64_m10_ss", "k64.fa", 64, 10, ["--SS"]),
    ("k96_m50", "k96.fa", 96, 50, []),
    ("k96_m20", "k96.fa", 96, 20, []),
    # unitigs of circular replicons, tandem repeats, hairpins and homopolymers (make_structured.py): contigs that overlap
    # themselves, their own reverse complement, and many others at once
    ("s_plasmids_k32_m0", "s_plasmids_k32.fa", 32, 0, []),
    ("s_tandem_k32_m5", "s_tandem_k32.fa", 32, 5, []),
    ("s_inverted_k40_m0_ss", "s_inverted_k40.fa", 40, 0, ["--SS"]),
    ("s_lowcomplex_k25_m0", "s_lowcomplex_k25.fa", 25, 0, []),
]
FORMATS = ["adj", "dot", "gfa1", "gfa2", "asqg", "sam"]


def main():
    out_dir = os.path.join(HERE, "adjlist")
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    for name, fa, k, m, extra in CASES:
        for fmt in FORMATS if name in ("k32_m0", "k64_m50", "k96_m20", "s_lowcomplex_k25_m0") else ["adj", "dot"]:
            r = subprocess.run([REF, "-k%d" % k, "-m%d" % m, "--" + fmt] + extra + [os.path.join(HERE, fa)],
                               stdout=subprocess.PIPE, check=True)
            # (the @PG line of SAM carries the command line of whoever ran it)
            data = b"".join(l for l in r.stdout.splitlines(True) if not l.startswith(b"@PG"))
            with open(os.path.join(out_dir, "%s.%s" % (name, fmt)), "wb") as f:
                f.write(data)
        index[name] = {"fasta": fa, "k": k, "m": m, "extra": extra}
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
        f.write("\n")


def full_size(fasta):
    path = os.path.join(HERE, "full_size.json")
    info = json.load(open(path))
    sha = hashlib.sha256(open(fasta, "rb").read()).hexdigest()
    runs = [r for r in info["runs"] if r["fasta_sha256"] == sha]
    if not runs:
        raise SystemExit("%s is none of the pinned full-size FASTAs" % fasta)
    k = runs[0]["k"]
    m = 50 if k > 50 else 0
    r = subprocess.run([REF, "-k%d" % k, "-m%d" % m, "--dot", fasta], stdout=subprocess.PIPE, check=True)
    runs[0]["adjlist"] = {"options": "-k%d -m%d --dot" % (k, m), "dot_sha256": hashlib.sha256(r.stdout).hexdigest(),
                          "dot_bytes": len(r.stdout), "edges": r.stdout.count(b" -> ")}
    json.dump(info, open(path, "w"), indent=1)
    print(json.dumps(runs[0]["adjlist"]))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--full-size":
        full_size(sys.argv[2])
    else:
        main()
