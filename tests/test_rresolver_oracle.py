"""The stage after AdjList in abyss-pe (bin/abyss-pe:581-585): abyss-rresolver-short.  SURVEY.md section 8 row f4.

What exists for it is the ORACLE: the unmodified RResolver/*.cpp of the reference compiled against
oracle/shim/btllib/ (a restatement of the btllib subset it uses -- btllib is not under /root/reference) into
oracle/_ref/abyss-rresolver-short, and golden runs of the whole rule on seeded read sets with short repeats
(tests/golden/make_rresolver.py).  These tests pin the oracle: the shim's ntHash / Bloom filter against their
definitions, and the oracle build against the committed goldens (same outputs at -j1 and -j4)."""
import json
import os
import subprocess

import numpy as np
import pytest

import rr_util
from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
RR = os.path.join(REF, "abyss-rresolver-short")
CHECK = os.path.join(REF, "btllib_check")
RRG = os.path.join(GOLDEN, "rresolver")

needs_ref = pytest.mark.skipif(not (os.path.exists(RR) and os.path.exists(CHECK)), reason="oracle/_ref not built (make -C oracle ref)")

SEED = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
M64 = (1 << 64) - 1


def _srol(x, n=1):
    for _ in range(n):
        m = ((x & 0x8000000000000000) >> 30) | ((x & 0x100000000) >> 32)
        x = ((x << 1) & 0xFFFFFFFDFFFFFFFF) | m
    return x


def _hashes(kmer, h):
    """ntHash of one k-mer from its definition (nthash.hpp:220-239 of the reference for the strand hashes; the canonical
    value of btllib's NtHash is their sum), and the multiply-shift extras (nthash.hpp:306-322)."""
    k = len(kmer)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    f = r = 0
    for i, c in enumerate(kmer):
        f ^= _srol(SEED[c], k - 1 - i)
        r ^= _srol(SEED[comp[c]], i)
    h0 = (f + r) & M64
    out = [h0]
    for i in range(1, h):
        t = (h0 * ((i ^ (k * 0x90b45d39fb6da1fa)) & M64)) & M64
        out.append(t ^ (t >> 27))
    return out


@needs_ref
def test_shim_nthash_is_the_definition_rolls_and_skips_non_acgt():
    rng = np.random.default_rng(2)
    for k, h in ((5, 3), (31, 7), (33, 2), (64, 7), (124, 7)):
        s = "".join(rng.choice(list("ACGT"), size=k + 60))
        s = s[:k + 10] + "N" + s[k + 11:k + 30] + "n" + s[k + 31:]  # two bad characters
        out = subprocess.run([CHECK, "hash", str(k), str(h), s], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        rows = {int(l.split()[0]): [int(x) for x in l.split()[1:]] for l in out if l}
        want = {p: _hashes(s[p:p + k].upper(), h) for p in range(len(s) - k + 1) if set(s[p:p + k].upper()) <= set("ACGT")}
        assert rows == want
        # strand symmetry: the reverse complement's k-mers hash alike
        fw = s.upper().replace("N", "T")
        rc = fw[::-1].translate(str.maketrans("ACGT", "TGCA"))
        a = subprocess.run([CHECK, "hash", str(k), str(h), fw], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        b = subprocess.run([CHECK, "hash", str(k), str(h), rc], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        assert [l.split()[1:] for l in a if l] == [l.split()[1:] for l in b if l][::-1]


@needs_ref
def test_shim_bloom_filter_layout_and_membership():
    rng = np.random.default_rng(3)
    k, h, nbytes = 20, 7, 4099  # (rounded up to 4104 bytes)
    ins = ["".join(rng.choice(list("ACGT"), size=60)) for _ in range(30)]
    qry = ins[:5] + ["".join(rng.choice(list("ACGT"), size=60)) for _ in range(5)]
    r = subprocess.run([CHECK, "bloom", str(nbytes), str(h), str(k)], input=("\n".join(ins) + "\n\n" + "\n".join(qry) + "\n").encode(),
                       stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    bits = bytearray(4104)
    for s in ins:
        for p in range(len(s) - k + 1):
            for v in _hashes(s[p:p + k], h):
                n = v % (4104 * 8)
                bits[n // 8] |= 1 << (n % 8)

    def found(s):
        return sum(all(bits[(v % (4104 * 8)) // 8] >> ((v % (4104 * 8)) % 8) & 1 for v in _hashes(s[p:p + k], h)) for p in range(len(s) - k + 1))
    assert [int(x) for x in r[:len(qry)]] == [found(s) for s in qry]
    assert [int(x) for x in r[:5]] == [41] * 5
    assert r[len(qry)].split() == [str(sum(bin(b).count("1") for b in bits)), "4104"]


@needs_ref
@pytest.mark.parametrize("name", rr_util.CASES)
def test_oracle_build_reproduces_the_rresolver_goldens(name, tmp_path):
    """make -C oracle ref && the rule of bin/abyss-pe:581-585 on the committed inputs: the resolved contigs, the resolved
    graph and every histogram are the committed ones, with one thread and with four."""
    info = rr_util.INDEX[name]
    want = rr_util.golden_outputs(name)
    assert any(f.endswith("-1-rr.fa") for f in want) and any(f.endswith("-1-rr.dot") for f in want) and len(want) >= 4
    for j in (1, 4):
        assert rr_util.run_case(RR, str(tmp_path), name, threads=j) == want, j
    assert info["contigs"] < info["unitigs"]  # (the stage did resolve repeats on this input)


@needs_ref
@pytest.mark.parametrize("v", rr_util.VARIANTS[::3], ids=rr_util.variant_id)
def test_oracle_build_reproduces_the_variant_digests(v, tmp_path):
    assert rr_util.run_variant(RR, str(tmp_path), v) == v["sha256"]
