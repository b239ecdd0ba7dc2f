"""AdjList on the GPU: abg_overlap_join through the C ABI and the drop-in
binary abyss_amd/bin/AdjList against the oracle restatement, the committed outputs of the
unmodified reference (tests/golden/adjlist, made by tests/golden/make_adjlist.py) and, where it
travelled with the repo, the reference binary itself (oracle/_ref/AdjList)."""
import os
import subprocess
import time

import numpy as np
import pytest

import adjlist_oracle as ao
from abyss_amd import api, build
from test_adjlist import FORMATS, INDEX, csr_of, golden_outputs, hc_join, run_bin, synthetic_contigs, write_fasta
from util import GOLDEN

pytestmark = pytest.mark.gpu

ADJLIST = os.path.join(build.BIN_DIR, "AdjList")


@pytest.fixture(scope="module")
def ov(_built):
    j = api.OverlapJoin()
    yield j
    j.close()


@pytest.mark.parametrize("name", sorted(INDEX))
def test_binary_writes_the_reference_outputs(name, _built):
    c = INDEX[name]
    for fmt, want in golden_outputs(name).items():
        assert run_bin(ADJLIST, c["k"], c["m"], fmt, c["extra"], os.path.join(GOLDEN, c["fasta"])) == want, (name, fmt)


@pytest.mark.parametrize("k,m,ss,seed", [(21, 0, False, 1), (33, 10, True, 3), (64, 50, False, 4), (65, 20, True, 5),
                                         (97, 50, False, 6), (130, 60, False, 7), (250, 200, False, 8)])
def test_binary_on_synthetic_contig_sets(k, m, ss, seed, tmp_path, _built):
    recs = synthetic_contigs(seed, k, n=3000, genome=400000)
    fa = str(tmp_path / "contigs.fa")
    write_fasta(fa, recs, width=70 if seed % 2 else 0)
    contigs, out = ao.build(recs, k, m, ss)
    extra = ["--SS"] if ss else []
    assert run_bin(ADJLIST, k, m, "adj", extra, fa) == ao.format_adj(contigs, out)
    assert run_bin(ADJLIST, k, m, "dot", extra, fa) == ao.format_dot(contigs, out)
    if os.path.exists(ao.REF_ADJLIST) and k - 1 <= 128:
        for fmt in FORMATS:
            assert run_bin(ADJLIST, k, m, fmt, extra, fa) == run_bin(ao.REF_ADJLIST, k, m, fmt, extra, fa), fmt


@pytest.mark.parametrize("overlap,ss", [(1, False), (2, True), (3, False), (31, False), (32, True), (33, False), (64, False),
                                        (65, True), (200, False), (256, False)])
def test_join_on_keys_that_collide(overlap, ss, ov):
    rng = np.random.default_rng(overlap)
    pool = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), overlap)) for _ in range(6)]
    pool += [ao.revcomp(pool[0]), b"A" * overlap, b"T" * overlap, (b"AT" * overlap)[:overlap]]
    seqs = [pool[int(rng.integers(len(pool)))] + b"C" + pool[int(rng.integers(len(pool)))] for _ in range(300)]
    contigs = ao.Contigs([("%d" % i, "", s) for i, s in enumerate(seqs)], overlap + 1)
    want_off, want_tgt = csr_of(ao.overlap_edges(contigs, ss))
    head, tail = api.pack_ends(seqs, overlap)
    off, tgt = ov.join(overlap, head, tail, ss)
    assert np.array_equal(off, want_off) and np.array_equal(tgt, want_tgt)


def test_join_of_nothing_one_and_again(ov):
    off, tgt = ov.join(5, np.zeros((0, 1), np.uint64), np.zeros((0, 1), np.uint64))
    assert off.tolist() == [0] and len(tgt) == 0
    head, tail = api.pack_ends([b"ACGTTTTTTACGT"], 4)
    off, tgt = ov.join(4, head, tail)
    want_off, want_tgt = csr_of(ao.overlap_edges(ao.Contigs([("0", "", b"ACGTTTTTTACGT")], 5)))
    assert off.tolist() == want_off.tolist() and tgt.tolist() == want_tgt.tolist()
    # bits past the key in its last word are the caller's business: they are ignored
    off2, tgt2 = ov.join(4, head | np.uint64(0xABCD << 8), tail | np.uint64(1 << 63))
    assert off2.tolist() == off.tolist() and tgt2.tolist() == tgt.tolist()


def test_join_rejects_what_it_cannot_do(ov):
    head, tail = api.pack_ends([b"ACGTACGT"], 4)
    for bad in (0, 257):
        with pytest.raises(api.AbyssAmdError):
            ov.join(bad, head, tail)


@pytest.mark.parametrize("overlap,n", [(63, 1_000_000), (95, 300_000)])
def test_join_of_millions_of_contigs_equals_the_serial_run_of_the_same_logic_and_the_oracle_on_a_slice(overlap, n, ov):
    # a de Bruijn-like set: n consecutive windows of a random genome, random orientation: ~2 edges a contig,
    # plus repeats (every 1000th window is a copy of window 0)
    rng = np.random.default_rng(overlap + n)
    step = 5
    genome = rng.integers(0, 4, n * step + overlap + 8, dtype=np.uint8)
    idx = np.arange(overlap)[None, :]
    starts = np.arange(n, dtype=np.int64) * step
    starts[::1000] = 0
    head_codes = genome[starts[:, None] + idx]
    tail_codes = genome[starts[:, None] + step + idx]
    flip = rng.random(n) < 0.5
    rc_head, rc_tail = 3 - tail_codes[:, ::-1], 3 - head_codes[:, ::-1]
    head_codes = np.where(flip[:, None], rc_head, head_codes)
    tail_codes = np.where(flip[:, None], rc_tail, tail_codes)
    W = (overlap + 31) // 32

    def pack(c):
        pad = np.zeros((n, W * 32), dtype=np.uint64)
        pad[:, :overlap] = c
        return np.ascontiguousarray((pad.reshape(n, W, 32) << (2 * np.arange(32, dtype=np.uint64))).sum(axis=2, dtype=np.uint64))
    head, tail = pack(head_codes), pack(tail_codes)
    ov.profile(True)
    t0 = time.time()
    off, tgt = ov.join(overlap, head, tail)
    wall = time.time() - t0
    print("\n[overlap join] %d contigs, k-1 = %d: %d edges in %.1f ms (kernels: %s)" % (
        n, overlap, len(tgt), wall * 1e3,
        ", ".join("%s %.2f" % (k, ov.profile_get(k)[0]) for k in ("overlap_keys", "sort_pairs", "overlap_count", "scan", "overlap_fill"))))
    ov.profile(False)
    want_off, want_tgt = hc_join(overlap, head, tail, False)
    assert np.array_equal(off, want_off) and np.array_equal(tgt, want_tgt)
    assert len(tgt) > 1.9 * n
    # the oracle (a dict of byte strings) on the first 20000 contigs, joined alone
    m = 20000
    seqs = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate([head_codes[i], [1], tail_codes[i]])]) for i in range(m)]
    want_off, want_tgt = csr_of(ao.overlap_edges(ao.Contigs([("%d" % i, "", s) for i, s in enumerate(seqs)], overlap + 1)))
    o2, t2 = ov.join(overlap, head[:m], tail[:m])
    assert np.array_equal(o2, want_off) and np.array_equal(t2, want_tgt)


def test_binary_errors_like_the_reference(tmp_path, _built):
    fa = str(tmp_path / "n.fa")
    write_fasta(fa, [("0", "", b"ACGTACGTNA")])
    r = subprocess.run([ADJLIST, "-k4", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"unexpected character: 'N'" in r.stderr
    write_fasta(fa, [("0", "", b"ACGTACGTTA")])
    r = subprocess.run([ADJLIST, "-k4", "--gpu=99", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"out of range" in r.stderr
    # (nothing to join, still no device: the device's verdict is not skipped)
    open(fa, "wb").close()
    r = subprocess.run([ADJLIST, "-k4", "--gpu=99", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"out of range" in r.stderr
