"""The drop-in `abyss-bloom-dbg` binary on a real MI355X: same command line as abyss-pe issues
(bin/abyss-pe:191-235,553-555: `abyss-bloom-dbg -k$k -q3 -b$B -j$j reads... > name-1.fa`),
byte-identical unitig FASTA / read log / trace against the reference's golden outputs and,
when the unmodified reference binary travelled with the snapshot (oracle/_ref), against a
live run of it on a bigger FASTQ read set (BASELINE.json configs[0] sized: 200 kbp, k=32)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import build, synth
from util import GoldenCase

pytestmark = pytest.mark.gpu


def cli():
    path = build.build_cli()
    assert path and os.path.exists(path)
    return path


def strip_length_column(trace: bytes) -> bytes:
    return b"".join(b"\t".join(r.split(b"\t")[:1] + r.split(b"\t")[2:]) + b"\n" for r in trace.splitlines())


@pytest.mark.parametrize("name", ["k32", "k64", "k25_h3_kc3_t40", "k40_mixed", "k48_K16", "k50_qr11", "s_plasmids_k32", "s_lowcomplex_k25", "s_inverted_k40"])
def test_cli_reproduces_golden(name, tmp_path):
    g = GoldenCase(name)
    fa = tmp_path / "reads.fa"
    with open(fa, "wb") as f:
        for i, s in enumerate(g.reads):
            f.write(b">r%d\n%s\n" % (i, s))
    r = subprocess.run([cli()] + g.meta["options"] + ["-j1", "-v", "--read-log=rl.tsv", "-T", "tr.tsv", "reads.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == g.fasta
    assert open(tmp_path / "rl.tsv", "rb").read() == g.readlog
    assert strip_length_column(open(tmp_path / "tr.tsv", "rb").read()) == g.trace
    assert ("popcount                = %d" % g.meta["filtered_popcount"]).encode() in r.stderr
    assert b"Assembly complete" in r.stderr


def test_cli_graphviz_dump(tmp_path):
    """`-g FILE` (bloom-dbg.cc:203-211): the GraphViz file against the one the reference wrote."""
    import hashlib
    import json
    from util import GOLDEN
    g = GoldenCase("k32")
    with open(tmp_path / "reads.fa", "wb") as f:
        for i, s in enumerate(g.reads):
            f.write(b">r%d\n%s\n" % (i, s))
    r = subprocess.run([cli()] + g.meta["options"] + ["-j4", "-v", "-g", "g.dot", "reads.fa"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == g.fasta
    ref = json.load(open(os.path.join(GOLDEN, "graph_golden.json")))["k32"]
    dot = open(tmp_path / "g.dot", "rb").read()
    assert len(dot) == ref["bytes"] and hashlib.sha256(dot).hexdigest() == ref["sha256"]
    assert ("(k-mers visited: %d, edges visited: %d)" % (ref["nodes"], ref["edges"])).encode() in r.stderr


def test_cli_partitioned_code_path_on_one_rank(tmp_path):
    """ABG_FORCE_DIST=1: the host binary creates the library's RCCL communicator (one rank), attaches
    it and runs the partitioned code path (what `--gpus N` does on every rank) -- same files as ever."""
    g = GoldenCase("k40_mixed")
    with open(tmp_path / "reads.fa", "wb") as f:  # FASTQ, parsed by 4 threads (-j4)
        for i, s in enumerate(g.reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))
    r = subprocess.run([cli()] + g.meta["options"] + ["-j4", "-v", "--gpus=1", "--read-log=rl.tsv", "-T", "tr.tsv", "reads.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ABG_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == g.fasta
    assert open(tmp_path / "rl.tsv", "rb").read() == g.readlog
    assert strip_length_column(open(tmp_path / "tr.tsv", "rb").read()) == g.trace
    assert ("popcount                = %d" % g.meta["filtered_popcount"]).encode() in r.stderr
    # more ranks than GPUs on the box: every rank must fail cleanly, not hang
    r = subprocess.run([cli()] + g.meta["options"] + ["--gpus=64", "reads.fa"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"invalid option" in r.stderr


def test_cli_option_errors(tmp_path):
    (tmp_path / "r.fa").write_text(">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    for args, msg in ((["-k32", "r.fa"], b"missing mandatory option `-b'"), (["-b1M", "r.fa"], b"missing mandatory option `-k'"),
                      (["-k32", "-b1M"], b"missing input file arguments"), (["-k32", "-b1M", "-K17", "r.fa"], b"must be <= k/2"),
                      (["-k32", "-b1M", "--qr-seed=7", "r.fa"], b"must be >= 11 and <= k/2"),
                      (["-k32", "-b1M", "-s101", "r.fa"], b"spaced seed must be exactly k bits long"),
                      (["-k32", "-bXYZ", "r.fa"], b"invalid option"),
                      (["-k32", "-b1M", "-C", "x.wig", "r.fa"], b"you must specify a reference with `-R' when using `-C'")):
        r = subprocess.run([cli()] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and msg in r.stderr, (args, r.stderr)
    r = subprocess.run([cli(), "-k32", "-b1M", "nonexistent.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"nonexistent.fq" in r.stderr


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/abyss-bloom-dbg did not travel with the snapshot")
def test_cli_matches_reference_binary_on_fastq(tmp_path):
    # BASELINE.json configs[0]: 200 kbp, 40x, 2x150 bp, k=32, B=100M, -q3, as abyss-pe runs it
    m1, m2 = synth.make_read_set(200000, 40.0)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)
    args = ["-k32", "-q3", "-b100M", "--read-log=rl_%s.tsv", "-T", "tr_%s.tsv", "r1.fq", "r2.fq"]
    ref_out, _ = ob.run_ref([a % "ref" if "%s" in a else a for a in args], cwd=str(tmp_path), threads=1)
    r = subprocess.run([cli(), "-j1"] + [a % "amd" if "%s" in a else a for a in args], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == ref_out
    assert len(ref_out) > 100000
    assert open(tmp_path / "rl_amd.tsv", "rb").read() == open(tmp_path / "rl_ref.tsv", "rb").read()
    assert strip_length_column(open(tmp_path / "tr_amd.tsv", "rb").read()) == \
        strip_length_column(open(tmp_path / "tr_ref.tsv", "rb").read())
    # gzip input goes through the same reader (Common/Uncompress.cpp in the reference)
    subprocess.run(["gzip", "-k", "r1.fq", "r2.fq"], cwd=tmp_path, check=True)
    rz = subprocess.run([cli(), "-k32", "-q3", "-b100M", "r1.fq.gz", "r2.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE)
    assert rz.returncode == 0 and rz.stdout == ref_out


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/abyss-bloom-dbg did not travel with the snapshot")
def test_cli_coverage_track_matches_reference_binary(tmp_path):
    """-C / -R: the 0/1 k-mer coverage WIG of a reference (writeCovTrack, bloom-dbg.h:1251-1334)."""
    h1, _ = synth.make_genome(60000, seed=11)
    m1, m2 = synth.make_read_set(60000, 7.0, genome_seed=11, read_seed=12)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)
    g = synth.codes_to_ascii(h1[None, :])[0].tobytes()
    with open(tmp_path / "ref.fa", "wb") as f:  # two records, lower case, a run of N, a short tail record
        f.write(b">chrA some comment\n" + g[:30000] + b"\n>chrB\n" + g[30000:45000].lower() + b"NNNNNNNNNN" + g[45000:] +
                b"\n>tiny\nACGTACGT\n")
    args = ["-k40", "-b8M", "-R", "ref.fa", "r1.fq", "r2.fq"]
    ref_out, _ = ob.run_ref(["-C", "cov_ref.wig"] + args, cwd=str(tmp_path), threads=1)
    r = subprocess.run([cli(), "-j1", "-C", "cov_amd.wig"] + args, cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == ref_out
    wig = open(tmp_path / "cov_ref.wig", "rb").read()
    assert wig.count(b"variableStep") > 10 and b"chrB" in wig
    assert open(tmp_path / "cov_amd.wig", "rb").read() == wig


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref/abyss-bloom-dbg did not travel with the snapshot")
def test_cli_checkpoints_match_reference_binary(tmp_path):
    """--checkpoint=N --keep-checkpoint: the four checkpoint files (BloomDBG/Checkpoint.h) after the last
    checkpoint, byte for byte; then a resumed run of ours continues from them."""
    g = GoldenCase("k32")
    with open(tmp_path / "reads.fa", "wb") as f:
        for i, s in enumerate(g.reads):
            f.write(b">r%d\n%s\n" % (i, s))
    args = ["-k32", "-b4M", "--checkpoint=1500", "--keep-checkpoint"]
    ref_out, _ = ob.run_ref(args + ["--checkpoint-prefix=ckr", "reads.fa"], cwd=str(tmp_path), threads=1)
    r = subprocess.run([cli(), "-j1", "-v"] + args + ["--checkpoint-prefix=cka", "reads.fa"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == ref_out == g.fasta
    assert b"Writing checkpoint data..." in r.stderr
    for ext in (".dbg.bloom", ".visited.bloom", ".counters.tsv", ".contigs.fa", ".contigs.fa.tmp"):
        assert open(tmp_path / ("cka" + ext), "rb").read() == open(tmp_path / ("ckr" + ext), "rb").read(), ext
    assert open(tmp_path / "cka.counters.tsv").read().split("\n")[1].split("\t")[1] == "3000"
    # resume: reads 0..2999 are skipped, their contigs come from the checkpoint, the rest is assembled
    r2 = subprocess.run([cli(), "-j1", "-v"] + args + ["--checkpoint-prefix=cka", "reads.fa"], cwd=tmp_path,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r2.returncode == 0, r2.stderr.decode()
    assert b"Resuming from last checkpoint..." in r2.stderr and b"Advancing to read index 3000" in r2.stderr
    assert r2.stdout == g.fasta
    # without --keep-checkpoint the files are removed at the end
    r3 = subprocess.run([cli(), "-j1", "-k32", "-b4M", "--checkpoint=1500", "--checkpoint-prefix=ckx", "reads.fa"], cwd=tmp_path,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r3.returncode == 0 and r3.stdout == g.fasta
    assert not [p for p in os.listdir(tmp_path) if p.startswith("ckx")]


def test_cli_reads_its_input_again_when_the_device_cannot_keep_the_reads(tmp_path):
    """abg_assemble_kept -> ABG_EAGAIN (forced: ABG_KEEP_FAIL): the binary falls back to the reference's
    second pass over the files and writes the same FASTA and read log."""
    m1, m2 = synth.make_read_set(150000, 30.0)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)
    outs = []
    # (ABG_KEEP_FAIL: the device store cannot grow; ABG_KEEP_LIMIT_BYTES: the ids the host keeps beside it outgrow its share of memory)
    for env in ({}, {"ABG_KEEP_FAIL": "1", "ABG_READER_WINDOW": "2000000"}, {"ABG_KEEP_LIMIT_BYTES": "100000", "ABG_READER_WINDOW": "2000000"}):
        r = subprocess.run([cli(), "-v", "-k40", "-q3", "-b64M", "-j8", "--read-log=rl.tsv", "r1.fq", "r2.fq"], cwd=tmp_path,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        outs.append((r.stdout, open(tmp_path / "rl.tsv", "rb").read(), r.stderr))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and outs[0][0].count(b">") > 100
    assert outs[0][0] == outs[2][0] and outs[0][1] == outs[2][1]
    assert b"reading the input again" in outs[1][2] and b"reading the input again" not in outs[0][2]
