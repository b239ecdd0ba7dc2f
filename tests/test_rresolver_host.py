"""abyss-rresolver-short (the rule after AdjList in Bloom mode, bin/abyss-pe:581-585) on the CPU: the product's
host side (abyss_amd/csrc/host/rresolver_core.h -- graph reader and surgery, read statistics, path support, writers) over the
product's read-filter logic (abyss_amd/csrc/abg_rr.h) run serially by tests/hostcheck/rresolver_check, against the runs of the
unmodified reference kept under tests/golden/rresolver.  The GPU twin is test_gpu_rresolver.py."""
import os
import subprocess

import numpy as np
import pytest

import rr_util
from abyss_amd import build

CHECK = os.path.join(os.path.dirname(build.HOSTCHECK), "rresolver_check")
REF_CHECK = os.path.join(build.ORACLE_DIR, "_ref", "btllib_check")


@pytest.mark.parametrize("name", rr_util.CASES)
def test_host_side_reproduces_the_reference_runs(name, tmp_path):
    """Contigs, graph and every histogram byte-equal to the reference's at -j1, whatever the number of reader threads."""
    want = rr_util.golden_outputs(name)
    for j in (1, 4):
        assert rr_util.run_case(CHECK, str(tmp_path), name, threads=j) == want, j


@pytest.mark.parametrize("v", rr_util.VARIANTS, ids=rr_util.variant_id)
def test_host_side_reproduces_the_option_variants(v, tmp_path):
    """Other formats in and out, -S / -U, explicit r values and factors, and few branching paths: with more than branching^2
    combinations the heads and tails are shuffled with rand(), so the repeats must be taken in the reference's -j1 order."""
    assert rr_util.run_variant(CHECK, str(tmp_path), v) == v["sha256"]


def test_small_reader_windows_change_nothing(tmp_path):
    """The reads cut into many windows and blocks by the parallel reader: the same filter, the same outputs."""
    env = dict(os.environ, ABG_READER_WINDOW="20000")
    assert rr_util.run_case(CHECK, str(tmp_path), "rr_mixed", threads=3, env=env) == rr_util.golden_outputs("rr_mixed")


@pytest.mark.parametrize("name", rr_util.CASES)
def test_contigs_read_block_parallel_change_nothing(name, tmp_path):
    """load_contigs over a plain FASTA file of a megabyte or more: blocks cut at record starts, a thread each (here forced on the
    small goldens, 2 .. 16 blocks): the same outputs."""
    want = rr_util.golden_outputs(name)
    for j in (2, 5, 16):
        env = dict(os.environ, ABG_FASTA_BLOCKS_MIN="1")
        assert rr_util.run_case(CHECK, str(tmp_path), name, threads=j, env=env) == want, j


@pytest.mark.skipif(not os.path.exists(REF_CHECK), reason="oracle/_ref not built (make -C oracle ref)")
def test_filter_logic_matches_the_btllib_restatement(tmp_path):
    """abg_rr.h run serially: the array after inserting read prefixes equals bit for bit what oracle/shim/btllib builds (its
    popcount and its answers to contains()), with N, lower case, reads shorter than r and a size that is not a multiple of 8."""
    rng = np.random.default_rng(12)
    r, h, nbytes, span = 31, 7, 70001, 36
    reads = []
    for i in range(400):
        s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(20, 90))).tobytes())
        if i % 7 == 0:
            s[int(rng.integers(0, len(s)))] = ord("N")
        if i % 5 == 0:
            s[3:9] = bytes(s[3:9]).lower()
        reads.append(bytes(s))
    qry = [x[:span] for x in reads[:40]] + [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 60).tobytes()) for _ in range(40)]
    out = subprocess.run([CHECK, "--filter", str(nbytes), str(h), str(r), str(span), str(tmp_path / "bits")], input=b"\n".join(reads) + b"\n",
                         stdout=subprocess.PIPE, check=True).stdout.split()
    ref = subprocess.run([REF_CHECK, "bloom", str(nbytes), str(h), str(r)], input=b"\n".join(x[:span] for x in reads) + b"\n\n" + b"\n".join(qry) + b"\n",
                         stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    assert [int(x) for x in out] == [int(x) for x in ref[len(qry)].split()]  # popcount, bytes (rounded up to 70008)
    # and from the definition: every bit of the array
    import test_rresolver_oracle as tro
    nb = int(out[1])
    bits = bytearray(nb)
    for s in reads:
        s = s[:span].upper()
        for p in range(len(s) - r + 1):
            if set(s[p:p + r]) <= set(b"ACGT"):
                for v in tro._hashes(s[p:p + r].decode(), h):
                    n = v % (nb * 8)
                    bits[n // 8] |= 1 << (n % 8)
    assert bytes(bits) == (tmp_path / "bits").read_bytes()


def test_cli_errors(tmp_path):
    def run(*args):
        return subprocess.run([CHECK] + list(args), cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    r = run("-k32")
    assert r.returncode == 1 and b"missing or invalid value for mandatory option `-b'" in r.stderr and b"missing input file arguments" in r.stderr
    r = run("-b1M", "-k32", "-g", "g", "-c", "c", "-m30", "-M20", "a", "b", "c")
    assert r.returncode == 1 and b"--min-tests cannot be higher than --max-tests" in r.stderr
    r = run("-b1Q", "-k32", "-g", "g", "-c", "c", "a", "b", "c")
    assert r.returncode == 1 and b"invalid option: `-b1Q'" in r.stderr
    r = run("-b1M", "-k32", "-e", "-g", "g", "-c", "c", "a", "b", "c")
    assert r.returncode == 1 and b"not supported" in r.stderr
    r = run("-b1M", "-k32", "-g", "g", "-c", "c", "missing.fa", "missing.dot", "missing.fq")
    assert r.returncode == 1 and b"missing.dot" in r.stderr
    r = run("--help")
    assert r.returncode == 0 and b"Usage: abyss-rresolver-short" in r.stdout
