"""abyss-pe plumbing (BASELINE.json configs[0]: "through `abyss-pe k=32 B=100M j=1` up to `-1.fa`").

The reference's pipeline driver is a Makefile (bin/abyss-pe); with `B=` set its `%-1.fa` rule is
`abyss-stack-size 65536 abyss-bloom-dbg $(abyssopt) $(in) $(se) > $@` (bin/abyss-pe:553-555,
191-235).  Two halves, because the reference tree exists only in the build container and a GPU
only on the GPU box:

* here (no GPU, reference present): the unmodified abyss-pe is run with OUR binary first on PATH,
  wrapped by a recorder; the recorded command line is the one the GPU tests feed the binary
  (tests/test_gpu_cli.py::test_cli_matches_reference_binary_on_fastq), our option parser accepts
  it (the run ends at "no HIP device", not at an option error), and the one-letter `name=` trap of
  SURVEY.md 8c (`name=t` injects `-t<files>`) is what it is;
* on the GPU box (`-m gpu`, no reference tree): a stand-in for that one rule -- the same command
  through an `abyss-stack-size` wrapper raising the stack limit -- writes the same `asm-1.fa` as
  the direct invocation.
"""
import os
import stat
import subprocess

import pytest

from abyss_amd import build, synth

REF_PE = "/root/reference/bin/abyss-pe"


def _write_reads(tmp_path, genome=20000, cov=10.0):
    m1, m2 = synth.make_read_set(genome, cov)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)


def _shim_dir(tmp_path, real):
    d = tmp_path / "shim"
    d.mkdir()
    rec = tmp_path / "argv.txt"
    sh = d / "abyss-bloom-dbg"
    sh.write_text("#!/bin/sh\nprintf '%%s\\n' \"$@\" > %s\nexec %s \"$@\"\n" % (rec, real))
    sh.chmod(sh.stat().st_mode | stat.S_IXUSR)
    return d, rec


@pytest.mark.skipif(not os.path.exists(REF_PE), reason="the reference tree is not on this machine")
def test_abyss_pe_issues_the_command_line_our_binary_accepts(tmp_path):
    cli = build.build_cli()
    _write_reads(tmp_path)
    shim, rec = _shim_dir(tmp_path, cli)
    env = dict(os.environ, PATH="%s:%s" % (shim, os.environ["PATH"]))
    r = subprocess.run(["make", "-rRf", REF_PE, "name=asm", "k=32", "B=100M", "j=1", "in=r1.fq r2.fq", "asm-1.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out = (r.stdout + r.stderr).decode()
    assert "abyss-stack-size 65536 abyss-bloom-dbg -k32 -q3" in out, out
    assert rec.read_text().split("\n")[:-1] == ["-k32", "-q3", "-b100M", "-j1", "r1.fq", "r2.fq"]
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        assert r.returncode == 0 and (tmp_path / "asm-1.fa").read_bytes().startswith(b">0 ")
    else:
        # the options were accepted and the inputs opened; what stops the run is the missing device
        assert r.returncode != 0 and "no HIP device" in out and "invalid option" not in out and "missing" not in out, out
    # the one-letter trap: `name=t` makes $(t) the input files, which abyss-pe turns into `-t<files>`
    r = subprocess.run(["make", "-rRf", REF_PE, "name=t", "k=32", "B=100M", "j=1", "in=r1.fq r2.fq", "t-1.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert any(a.startswith("-t") and "r1.fq" in a for a in rec.read_text().split("\n")), rec.read_text()


@pytest.mark.gpu
def test_pipeline_rule_writes_the_same_file_as_the_direct_call(tmp_path):
    cli = build.build_cli()
    _write_reads(tmp_path, genome=200000, cov=40.0)
    wrap = tmp_path / "abyss-stack-size"  # what bin/abyss-stack-size does: raise the soft stack limit, run the command
    wrap.write_text("#!/bin/sh\nulimit -s \"$1\" 2>/dev/null\nshift\nexec \"$@\"\n")
    wrap.chmod(wrap.stat().st_mode | stat.S_IXUSR)
    argv = ["-k32", "-q3", "-b100M", "-j1", "r1.fq", "r2.fq"]
    direct = subprocess.run([cli] + argv, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert direct.returncode == 0, direct.stderr.decode()
    env = dict(os.environ, PATH="%s:%s:%s" % (tmp_path, os.path.dirname(cli), os.environ["PATH"]))
    rule = "abyss-stack-size 65536 abyss-bloom-dbg %s > asm-1.fa" % " ".join(argv)
    r = subprocess.run(["sh", "-c", rule], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()
    assert (tmp_path / "asm-1.fa").read_bytes() == direct.stdout and direct.stdout.count(b">") > 300


@pytest.mark.skipif(not os.path.exists(REF_PE), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("k,extra,want", [(32, [], ["-k32", "-m0", "--dot", "asm-1.fa"]),
                                           (64, ["SS=--SS"], ["--SS", "-k64", "-m50", "--dot", "asm-1.fa"]),
                                           (96, ["v=-v", "graph=gfa2"], ["-v", "-k96", "-m50", "--gfa2", "asm-1.fa"])])
def test_abyss_pe_issues_the_adjlist_command_line_our_binary_accepts(tmp_path, k, extra, want):
    """The step after `-1.fa` (bin/abyss-pe:238-246,575-577): `AdjList $(alopt) --$g asm-1.fa >asm-1.$g`."""
    build.build_cli()
    real = os.path.join(build.BIN_DIR, "AdjList")
    d = tmp_path / "shim"
    d.mkdir()
    rec = tmp_path / "argv.txt"
    sh = d / "AdjList"
    sh.write_text("#!/bin/sh\nprintf '%%s\\n' \"$@\" > %s\nexec %s \"$@\"\n" % (rec, real))
    sh.chmod(sh.stat().st_mode | stat.S_IXUSR)
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "k%d.fa" % k)
    (tmp_path / "asm-1.fa").write_bytes(open(golden, "rb").read())
    g = "gfa2" if "graph=gfa2" in extra else "dot"
    env = dict(os.environ, PATH="%s:%s" % (d, os.environ["PATH"]))
    r = subprocess.run(["make", "-rRf", REF_PE, "name=asm", "k=%d" % k, "B=100M", "j=1", "in=r1.fq r2.fq"] + extra + ["asm-1.%s" % g],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out = (r.stdout + r.stderr).decode()
    assert rec.exists(), out
    assert rec.read_text().split("\n")[:-1] == want
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        assert r.returncode == 0, out
    else:
        assert r.returncode != 0 and "no HIP device" in out and "invalid option" not in out, out


@pytest.mark.gpu
def test_the_two_rules_in_a_row_write_the_reference_overlap_graph(tmp_path):
    """`%-1.fa` then `%-1.dot` as abyss-pe issues them (bin/abyss-pe:553-555,575-577), both drop-ins from
    PATH; the graph is what the reference's AdjList (or, without it, its restatement) makes of the unitigs."""
    import adjlist_oracle as ao
    cli = build.build_cli()
    _write_reads(tmp_path, genome=120000, cov=30.0)
    env = dict(os.environ, PATH="%s:%s" % (os.path.dirname(cli), os.environ["PATH"]))
    for rule in ("abyss-bloom-dbg -k40 -q3 -b64M -j4 r1.fq r2.fq > asm-1.fa", "AdjList -k40 -m0 --dot asm-1.fa > asm-1.dot",
                 "AdjList -k40 -m20 --adj asm-1.fa > asm-1.adj"):
        r = subprocess.run(["sh", "-c", rule], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
    recs = ao.read_fasta(str(tmp_path / "asm-1.fa"))
    assert len(recs) > 100
    contigs, out = ao.build(recs, 40, 0)
    assert (tmp_path / "asm-1.dot").read_bytes() == ao.format_dot(contigs, out)
    contigs, out = ao.build(recs, 40, 20)
    assert (tmp_path / "asm-1.adj").read_bytes() == ao.format_adj(contigs, out)
    if os.path.exists(ao.REF_ADJLIST):
        r = subprocess.run([ao.REF_ADJLIST, "-k40", "-m0", "--dot", "asm-1.fa"], cwd=tmp_path, stdout=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == (tmp_path / "asm-1.dot").read_bytes()


@pytest.mark.skipif(not os.path.exists(REF_PE), reason="the reference tree is not on this machine")
def test_abyss_pe_issues_the_rresolver_command_line_our_binary_accepts(tmp_path):
    """The rule after `-1.dot` in Bloom mode (bin/abyss-pe:581-585): `abyss-rresolver-short -b$B -f0.8 -j$j -k$k -h asm-1-rr --dot
    -c asm-1-rr.fa -g asm-1-rr.dot asm-1.fa asm-1.dot $(in)`."""
    import rr_util
    build.build_cli()
    real = os.path.join(build.BIN_DIR, "abyss-rresolver-short")
    d = tmp_path / "shim"
    d.mkdir()
    rec = tmp_path / "argv.txt"
    sh = d / "abyss-rresolver-short"
    sh.write_text("#!/bin/sh\nprintf '%%s\\n' \"$@\" > %s\nexec %s \"$@\"\n" % (rec, real))
    sh.chmod(sh.stat().st_mode | stat.S_IXUSR)
    reads = rr_util.write_inputs(str(tmp_path), "rr_k32")
    os.rename(tmp_path / "rr_k32-1.fa", tmp_path / "asm-1.fa")
    os.rename(tmp_path / "rr_k32-1.dot", tmp_path / "asm-1.dot")
    env = dict(os.environ, PATH="%s:%s" % (d, os.environ["PATH"]))
    r = subprocess.run(["make", "-rRf", REF_PE, "name=asm", "k=32", "B=8M", "j=2", "in=%s" % " ".join(reads), "asm-1-rr.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out = (r.stdout + r.stderr).decode()
    assert rec.exists(), out
    assert rec.read_text().split("\n")[:-1] == ["-b8M", "-f0.8", "-j2", "-k32", "-h", "asm-1-rr", "--dot", "-c", "asm-1-rr.fa", "-g", "asm-1-rr.dot",
                                                 "asm-1.fa", "asm-1.dot"] + reads
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        assert r.returncode == 0, out
        assert (tmp_path / "asm-1-rr.fa").read_bytes() == rr_util.golden_outputs("rr_k32")["rr_k32-1-rr.fa"]
    else:
        assert r.returncode != 0 and "no HIP device" in out and "invalid option" not in out and "missing" not in out, out


@pytest.mark.gpu
def test_the_three_rules_in_a_row_write_what_the_reference_binaries_write(tmp_path):
    """`%-1.fa`, `%-1.dot`, `%-1-rr.fa %-1-rr.dot` as abyss-pe issues them in Bloom mode (bin/abyss-pe:553-555,575-577,581-585), the
    three drop-ins from PATH on a read set with short repeats; every file equals what the unmodified reference binaries
    (oracle/_ref, -j1) write on the same reads."""
    import numpy as np
    import rr_util
    cli = build.build_cli()
    name = "rr_k64"
    reads = rr_util.write_inputs(str(tmp_path), name)
    env = dict(os.environ, PATH="%s:%s" % (os.path.dirname(cli), os.environ["PATH"]))
    rules = ["abyss-bloom-dbg -k64 -b16M -j2 %s > asm-1.fa" % " ".join(reads),
             "AdjList -k64 -m50 --dot asm-1.fa > asm-1.dot",
             "abyss-rresolver-short -b16M -f0.8 -j2 -k64 -h asm-1-rr --dot -c asm-1-rr.fa -g asm-1-rr.dot asm-1.fa asm-1.dot %s" % " ".join(reads)]
    for rule in rules:
        r = subprocess.run(["sh", "-c", rule], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
    # the golden case was made by the same three rules with the reference's binaries (tests/golden/make_rresolver.py)
    assert (tmp_path / "asm-1.fa").read_bytes() == open(os.path.join(rr_util.RRG, name + "-1.fa"), "rb").read()
    assert (tmp_path / "asm-1.dot").read_bytes() == open(os.path.join(rr_util.RRG, name + "-1.dot"), "rb").read()
    want = rr_util.golden_outputs(name)
    for f, data in want.items():
        assert (tmp_path / f.replace(name, "asm")).read_bytes() == data, f
