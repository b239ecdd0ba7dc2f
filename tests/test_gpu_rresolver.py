"""abyss-rresolver-short on the GPU (bin/abyss-pe:581-585): the read filter's kernels through the C ABI
(abg_rr_*) against the filter the reference's own classes build (oracle/_ref/btllib_check: the btllib restatement the oracle
binary links, see oracle/shim/btllib), and the drop-in binary against the reference's runs under tests/golden/rresolver."""
import os
import subprocess

import numpy as np
import pytest

import rr_util
from abyss_amd import api, build

pytestmark = pytest.mark.gpu

REF_CHECK = os.path.join(build.ORACLE_DIR, "_ref", "btllib_check")
EXE = os.path.join(build.BIN_DIR, "abyss-rresolver-short")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_CHECK), reason="oracle/_ref not built (make -C oracle ref)")


def reference_filter(tmp_path, reads, nbytes, h, r, span):
    out = subprocess.run([REF_CHECK, "bloomdump", str(nbytes), str(h), str(r), str(span), str(tmp_path / "ref.bits")],
                         input=b"\n".join(reads) + b"\n", stdout=subprocess.PIPE, check=True).stdout.split()
    return np.fromfile(tmp_path / "ref.bits", dtype=np.uint8), int(out[0]), int(out[1])


def reference_counts(reads_in, queries, nbytes, h, r):
    out = subprocess.run([REF_CHECK, "bloom", str(nbytes), str(h), str(r)], input=b"\n".join(reads_in) + b"\n\n" + b"\n".join(queries) + b"\n",
                         stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    return [int(x) for x in out[:len(queries)]]


def to_buf(seqs):
    return b"".join(seqs), np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)


def messy_reads(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(rng.integers(lo, hi))).tobytes())
        if i % 7 == 0:
            s[int(rng.integers(0, len(s)))] = ord("N")
        if i % 11 == 0:
            s[int(rng.integers(0, len(s)))] = ord("n")
        if i % 5 == 0:
            s[3:9] = bytes(s[3:9]).lower()
        out.append(bytes(s))
    return out


@needs_ref
@pytest.mark.parametrize("r,span,nbytes", [(31, 36, 70001), (124, 127, 1 << 20), (92, 95, 3 << 19), (200, 230, 1 << 18), (300, 420, 1 << 18), (1100, 1200, 1 << 16)])
def test_filter_array_is_the_reference_filters(tmp_path, r, span, nbytes):
    """Every bit: reads with N / n / lower case, reads shorter than r and than the span, a size that is not a multiple of 8,
    and record lengths that take each of the insert kernel's variants (256, 128, 64 lanes a workgroup; records read in place)."""
    reads = messy_reads(3000, max(8, r - 20), span + 60, 5 + r)
    want, pop, nb = reference_filter(tmp_path, reads, nbytes, 7, r, span)
    f = api.ReadFilter(nbytes, r)
    assert f.nbytes == nb
    buf, off = to_buf(reads)
    assert f.insert(buf, off, span) == len(reads)
    assert f.popcount() == pop
    assert np.array_equal(f.export(), want)
    # inserting again changes nothing; clear() empties it
    f.insert(buf, off, span)
    assert np.array_equal(f.export(), want)
    f.clear()
    assert f.popcount() == 0
    f.close()


@needs_ref
def test_length_filter_and_counts(tmp_path):
    """Only the reads of the wanted lengths go in (BloomFilters.cpp:182), and contains() counts what the reference's counts."""
    r, span, nbytes = 40, 43, 1 << 17
    reads = messy_reads(2000, 30, 70, 3)
    wanted = [50, 51, 60]
    chosen = [s for s in reads if len(s) in wanted]
    want, pop, _ = reference_filter(tmp_path, chosen, nbytes, 7, r, span)
    f = api.ReadFilter(nbytes, r)
    buf, off = to_buf(reads)
    assert f.insert(buf, off, span, wanted) == len(chosen)
    assert np.array_equal(f.export(), want)
    # queries: inserted prefixes (all of their r-mers), other reads, sequences with N, shorter than r, and long ones (> 64 r-mers)
    rng = np.random.default_rng(8)
    qry = [s[:span] for s in chosen[:50]] + reads[:200] + [b"ACGT" * 5] + [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 400).tobytes())]
    qry += [chosen[0][:span] + chosen[1][:span] + b"N" + chosen[2][:span]]
    got = f.contains(*to_buf(qry))
    assert [int(x) for x in got] == reference_counts([s[:span] for s in chosen if len(s[:span]) >= r], qry, nbytes, 7, r)
    assert int(got[0]) == span - r + 1
    assert len(f.contains(b"", np.zeros(1, dtype=np.uint64))) == 0
    f.close()


@needs_ref
@pytest.mark.parametrize("name", rr_util.CASES)
def test_filter_of_a_golden_read_set(tmp_path, name):
    info = rr_util.INDEX[name]
    d = np.load(os.path.join(rr_util.RRG, name + ".reads.npz"))
    buf, off = d["buf"].tobytes(), d["off"].astype(np.uint64)
    reads = [buf[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    r = min(info["k"] + 60, max(len(s) for s in reads) - 4 + 1)
    span, nbytes = r + 3, int(0.8 * (8 << 20))
    want, pop, _ = reference_filter(tmp_path, reads, nbytes, 7, r, span)
    f = api.ReadFilter(nbytes, r)
    f.profile(True)
    f.insert(buf, off, span)
    assert np.array_equal(f.export(), want) and f.popcount() == pop
    assert f.profile_get("rr_insert")[1] >= 1
    f.close()


@pytest.mark.parametrize("name", rr_util.CASES)
def test_drop_in_binary_reproduces_the_reference_runs(name, tmp_path):
    want = rr_util.golden_outputs(name)
    for j in (1, 4):
        assert rr_util.run_case(EXE, str(tmp_path), name, threads=j) == want, j


@pytest.mark.parametrize("v", rr_util.VARIANTS, ids=rr_util.variant_id)
def test_drop_in_binary_reproduces_the_option_variants(v, tmp_path):
    assert rr_util.run_variant(EXE, str(tmp_path), v, threads=2) == v["sha256"]


def test_many_staging_slots_and_small_windows(tmp_path):
    """More reads than a staging slot holds and a reader cut into small windows: same outputs."""
    env = dict(os.environ, ABG_READER_WINDOW="30000", ABG_RR_PACK_THREADS="3")
    assert rr_util.run_case(EXE, str(tmp_path), "rr_mixed", threads=3, env=env) == rr_util.golden_outputs("rr_mixed")
    # 600 k prefixes of 128 bytes = 2.3 slots of 32 MB
    rng = np.random.default_rng(1)
    base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (1000, 140))
    reads = [bytes(base[i % 1000]) for i in range(600000)]
    f = api.ReadFilter(1 << 20, 124)
    f.insert(*to_buf(reads), 127)
    g = api.ReadFilter(1 << 20, 124)
    g.insert(*to_buf(reads[:1000]), 127)
    assert np.array_equal(f.export(), g.export())
    f.close()
    g.close()
