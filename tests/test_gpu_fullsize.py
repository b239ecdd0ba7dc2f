"""Properties of the HIP path at BASELINE.json's full size (configs[1]: 5 M x 2x150 bp, k=64, B=2G,
H=4), where the oracle cannot follow.  The path is a deterministic restatement of a sequential
algorithm, so its results must not depend on how the work was cut up:

* the counting filter, the visited filter, the assembly counters and every unitig are identical
  whether PASS 1 runs in 2^24- or 2^22-k-mer batches against a large or a small claim table, PASS 2
  in the default batch schedule or another one, the commit as the parallel fixed point or as the
  ordered single-workgroup loop;
* assembling the same reads again yields nothing (every read is then entirely visited);
* PASS 1 has no false negatives: every k-mer of a sample of reads has min count >= 1, and every
  k-mer of every unitig is solid (min count >= kc) and visited.

Exactness against the oracle / the reference binary is tested at sizes they finish in seconds
(test_gpu_parity.py, test_gpu_cli.py); this file ties the full-size runs to those.
"""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from abyss_amd import api  # noqa: E402

pytestmark = pytest.mark.gpu

K, BLOOM, PAIRS, READ_LEN = 64, 2 << 30, 5_000_000, 150


def _run(words, woff, lens, n_reads, env, monkeypatch, want_contigs, comm=None, **tuning):
    for key in ("ABG_PAR_COMMIT", "ABG_P2_FIRST_BATCH", "ABG_P2_MAX_BATCH", "ABG_DRAIN_THRESHOLD", "ABG_FORCE_DIST"):
        monkeypatch.delenv(key, raising=False)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    g = api.BloomDBG(K, bloom_bytes=BLOOM, num_hashes=4, min_cov=2, **tuning)
    if comm is not None:
        g.attach_comm(comm)
    g.load_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), n_reads)
    counters = g.counters()
    _, contigs = g.assemble_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), n_reads, want_results=False,
                                   want_contigs=want_contigs)
    digest = hashlib.sha1()
    for c in contigs:
        if not c.redundant:
            digest.update(b"%d %d %d %d " % (c.contig_id, c.read_index, len(c.seq), c.coverage))
            digest.update(c.seq)
    return g, counters, contigs, digest.hexdigest()


def test_full_size_results_do_not_depend_on_the_execution_schedule(monkeypatch):
    import torch
    import bench
    device = torch.device("cuda:0")
    genome_len = int(PAIRS * 2 * READ_LEN / 50.0)
    words, woff, lens = bench.gen_packed_reads(genome_len, PAIRS, READ_LEN, 0.005, seed=42, device=device)
    n_reads = 2 * PAIRS
    torch.cuda.synchronize()

    a, cnt_a, contigs_a, dig_a = _run(words, woff, lens, n_reads, {}, monkeypatch, True)
    ca, sa = a.assembly_counters(), a.stats()
    # byte for byte the reference's -j1 FASTA on this read set (tests/golden/full_size.json: the unmodified
    # reference run on the FASTQ files of synth.make_read_set_cb, whose torch twin generated the reads above)
    golden = bench.golden_for(1, PAIRS, K, 0, "2G")
    assert golden is not None, "tests/golden/full_size.json has no run for configs[1]"
    fasta = hashlib.sha256()
    n_unitigs = 0
    for c in contigs_a:
        if not c.redundant:
            mate = 1 if c.read_index < PAIRS else 2
            fasta.update(b">%d %d %d read:r%d/%d\n%s\n" % (c.contig_id, len(c.seq), c.coverage, c.read_index % PAIRS, mate, c.seq))
            n_unitigs += 1
    assert (n_unitigs, ca["bases_assembled"]) == (golden["unitigs"], golden["unitig_bp"])
    assert fasta.hexdigest() == golden["fasta_sha256"]
    assert a.counting_stats()[1] == golden["filtered_popcount"]
    # ... and abyss-pe's next step on those unitigs (AdjList -k64 -m50 --dot, bin/abyss-pe:238-246,575-577):
    # the drop-in AdjList writes the overlap graph the reference AdjList wrote for the reference's FASTA
    if "adjlist" in golden:
        import subprocess
        import tempfile
        from abyss_amd import build
        with tempfile.TemporaryDirectory() as td:
            fa = os.path.join(td, "unitigs-1.fa")
            with open(fa, "wb") as f:
                for c in contigs_a:
                    if not c.redundant:
                        mate = 1 if c.read_index < PAIRS else 2
                        f.write(b">%d %d %d read:r%d/%d\n%s\n" % (c.contig_id, len(c.seq), c.coverage, c.read_index % PAIRS, mate, c.seq))
            r = subprocess.run([os.path.join(build.BIN_DIR, "AdjList")] + golden["adjlist"]["options"].split() + [fa],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()
            assert r.stdout.count(b" -> ") == golden["adjlist"]["edges"]
            assert hashlib.sha256(r.stdout).hexdigest() == golden["adjlist"]["dot_sha256"]
    # the searches went through the chain shortcuts and the memo
    assert sa["chain_steps"] > 0 and sa["memo_hits"] > 0
    vis_a = a.visited()
    assert sa["commit_rounds"] > 0 and ca["next_contig_id"] > 50_000

    # PASS 1 sanity on a sample: no false negatives (every read k-mer was inserted at least once)
    nonred = [c for c in contigs_a if not c.redundant]
    assert len(nonred) == ca["next_contig_id"] and sum(len(c.seq) for c in nonred) == ca["bases_assembled"]
    # every unitig k-mer is solid and, after the run, visited: assembling the unitigs themselves as
    # reads finds every one of them ALL_KMERS_VISITED (result 5) or rejects it before (never "NOT_SOLID")
    sample = nonred[:: max(1, len(nonred) // 2000)]
    buf, off = api.concat_seqs([c.seq for c in sample])
    res, new_contigs = a.assemble(buf, off)
    assert set(np.unique(res)) <= {3, 5}, np.bincount(res)  # BLUNT_END (contig ends at a tip) or ALL_KMERS_VISITED
    assert not [c for c in new_contigs if not c.redundant]

    # idempotence: the same reads again -> nothing new, all solid reads are visited
    before = a.assembly_counters()
    _, again = a.assemble_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), n_reads, want_results=False)
    after = a.assembly_counters()
    assert not [c for c in again if not c.redundant]
    assert after["next_contig_id"] == before["next_contig_id"] and after["bases_assembled"] == before["bases_assembled"]
    assert np.array_equal(a.visited(), vis_a)
    a.close()

    # another way of cutting up the same work: small PASS 1 batches and claim table, another PASS 2
    # schedule, fewer walkers in flight, the ordered commit kernel
    b, cnt_b, contigs_b, dig_b = _run(words, woff, lens, n_reads,
                                      {"ABG_PAR_COMMIT": "0", "ABG_P2_FIRST_BATCH": "50000", "ABG_P2_MAX_BATCH": "1500000",
                                       "ABG_DRAIN_THRESHOLD": "100000"},
                                      monkeypatch, True, insert_batch_kmers=1 << 22, claim_log2=26, walk_slots=1024)
    assert b.stats()["commit_rounds"] == 0
    assert np.array_equal(cnt_a, cnt_b)
    cb = b.assembly_counters()
    assert {k2: cb[k2] for k2 in ca} == ca
    assert dig_a == dig_b
    assert np.array_equal(vis_a, b.visited())
    b.close()

    # and the partitioned code path of the multi-GPU run on one rank: evaluate /
    # all_reduce / apply rounds with the compacted loser lists, the drain hand-over, split classification,
    # merged walk results -- every collective an identity (abyss_amd.dist.LocalComm)
    from abyss_amd import dist as adist
    comm = adist.LocalComm()
    c, cnt_c, contigs_c, dig_c = _run(words, woff, lens, n_reads, {"ABG_FORCE_DIST": "1"}, monkeypatch, True, comm=comm)
    assert comm.calls["all_reduce"] > 100 and comm.calls["all_gather_v"] > 0
    assert np.array_equal(cnt_a, cnt_c)
    cc = c.assembly_counters()
    assert {k2: cc[k2] for k2 in ca} == ca
    assert dig_a == dig_c
    assert np.array_equal(vis_a, c.visited())
    c.close()


def test_full_size_spaced_seed_run_is_the_reference_fasta():
    """BASELINE.json configs[3] (-k96 -K32, 5 M pairs, B=2G) at its stated size: the unitig FASTA through the
    C ABI, byte for byte the reference's -j1 output on the same read set (tests/golden/full_size.json)."""
    import torch
    import bench
    device = torch.device("cuda:0")
    genome_len = int(PAIRS * 2 * READ_LEN / 50.0)
    words, woff, lens = bench.gen_packed_reads(genome_len, PAIRS, READ_LEN, 0.005, seed=42, device=device)
    torch.cuda.synchronize()
    golden = bench.golden_for(3, PAIRS, 96, 32, "2G")
    assert golden is not None, "tests/golden/full_size.json has no run for configs[3]"
    g = api.BloomDBG(96, bloom_bytes=BLOOM, num_hashes=4, min_cov=2, spaced_seed=api.spaced_seed_kmer_pair(96, 32))
    g.load_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), 2 * PAIRS)
    assert g.counting_stats()[1] == golden["filtered_popcount"]
    _, contigs = g.assemble_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), 2 * PAIRS, want_results=False, want_contigs=True)
    fasta = hashlib.sha256()
    n = bp = 0
    for c in contigs:
        if not c.redundant:
            fasta.update(b">%d %d %d read:r%d/%d\n%s\n" % (c.contig_id, len(c.seq), c.coverage, c.read_index % PAIRS, 1 if c.read_index < PAIRS else 2, c.seq))
            n += 1
            bp += len(c.seq)
    assert (n, bp) == (golden["unitigs"], golden["unitig_bp"])
    assert fasta.hexdigest() == golden["fasta_sha256"]
    g.close()


def test_full_size_run_in_configs2_regime_is_the_reference_fasta():
    """configs[2] gives its counting filter 35.8 bytes per genome base (B=40G for 1.2 Gbp), half of configs[1]'s: the false-positive
    branches, the crowded counters of PASS 1 and the commit's redundancy tests all run at another rate.  Its reference run at size
    is beyond a round of this build (14-15 h at -j1, 46 GB), so the regime is pinned at configs[1]'s size: the 5 M pairs with -b1G
    (1 GiB / 30 Mbp), the unmodified reference at -j1 (tests/golden/make_full_size.py --only ref_c2regime).  Same bytes."""
    import torch
    import bench
    device = torch.device("cuda:0")
    genome_len = int(PAIRS * 2 * READ_LEN / 50.0)
    words, woff, lens = bench.gen_packed_reads(genome_len, PAIRS, READ_LEN, 0.005, seed=42, device=device)
    torch.cuda.synchronize()
    golden = bench.golden_for(2, PAIRS, K, 0, "1G")
    assert golden is not None, "tests/golden/full_size.json has no run in configs[2]'s regime"
    g = api.BloomDBG(K, bloom_bytes=1 << 30, num_hashes=4, min_cov=2)
    g.load_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), 2 * PAIRS)
    assert g.counting_stats()[1] == golden["filtered_popcount"]
    _, contigs = g.assemble_packed(words.data_ptr(), woff.data_ptr(), lens.data_ptr(), 2 * PAIRS, want_results=False, want_contigs=True)
    fasta = hashlib.sha256()
    n = bp = 0
    for c in contigs:
        if not c.redundant:
            fasta.update(b">%d %d %d read:r%d/%d\n%s\n" % (c.contig_id, len(c.seq), c.coverage, c.read_index % PAIRS, 1 if c.read_index < PAIRS else 2, c.seq))
            n += 1
            bp += len(c.seq)
    assert (n, bp) == (golden["unitigs"], golden["unitig_bp"])
    assert fasta.hexdigest() == golden["fasta_sha256"]
    ca = g.assembly_counters()
    assert "solid reads: %d" % ca["solid_reads"] in golden["last_progress_line"] and "visited reads: %d" % ca["visited_reads"] in golden["last_progress_line"]
    g.close()
