"""The drop-in binary's sequence reader (abyss_amd/csrc/host/fasta_reader.h) against the
reference's own FastaReader: golden record dumps made with oracle/_ref/ref_reader
(tests/golden/reader_*.tsv) and, where the reference build is present, a live comparison."""
import os
import subprocess

import pytest

import oracle_binding as ob
from abyss_amd import build
from util import GOLDEN

OPTS = [[], ["-q", "3"], ["-q", "3", "-Q", "10"], ["--no-chastity"], ["--no-trim-masked", "-q", "20"]]
REF_READER = os.path.join(os.path.dirname(ob.REF_BIN), "ref_reader")


def run_mine(opts, path):
    build.build_hostcheck()
    return subprocess.run([build.READER_CHECK] + opts + [path], stdout=subprocess.PIPE, check=True).stdout


@pytest.mark.parametrize("i", range(len(OPTS)))
def test_reader_matches_golden(i):
    inp = os.path.join(GOLDEN, "reader_input.fq")
    assert run_mine(OPTS[i], inp) == open(os.path.join(GOLDEN, "reader_%d.tsv" % i), "rb").read()


@pytest.mark.parametrize("tag,name", [("sam_", "reader_input.sam"), ("qseq_", "reader_input_qseq.txt"),
                                      ("export_", "reader_input_export.txt")])
@pytest.mark.parametrize("i", range(len(OPTS)))
def test_tabular_formats_match_golden(tag, name, i):
    inp = os.path.join(GOLDEN, name)
    want = open(os.path.join(GOLDEN, "reader_%s%d.tsv" % (tag, i)), "rb").read()
    assert run_mine(OPTS[i], inp) == want
    assert run_mine(OPTS[i] + ["-j", "4"], inp) == want  # not FASTQ (a SAM header begins with '@' too): sequential reader


@pytest.mark.skipif(not os.path.exists(REF_READER), reason="reference build not present")
def test_reader_matches_reference_live(tmp_path):
    import numpy as np
    rng = np.random.default_rng(5)
    lines = []
    for i in range(400):
        L = int(rng.integers(1, 120))
        seq = "".join(rng.choice(list("ACGTacgtN"), p=[.22, .22, .22, .22, .02, .02, .02, .02, .04], size=L))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 74, size=L))
        if qual.startswith("@"):  # a quality line starting with '@' is legal but confusing; avoid it in the fixture
            qual = "I" + qual[1:]
        casava = ["", " 1:N:0:ACGT", " 2:Y:0:ACGT"][int(rng.integers(0, 3))]
        lines.append("@read%d%s\n%s\n+\n%s\n" % (i, casava, seq, qual))
    p = tmp_path / "x.fq"
    p.write_text("".join(lines))
    for opts in OPTS + [["-q", "15", "-Q", "20", "--no-chastity"]]:
        ref = subprocess.run([REF_READER] + opts + [str(p)], stdout=subprocess.PIPE, check=True).stdout
        assert run_mine(opts, str(p)) == ref, opts


def _sam_and_qseq(tmp_path, seed=9):
    import numpy as np
    rng = np.random.default_rng(seed)

    def seq_qual(L, lo=33, hi=74):
        seq = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=L))
        qual = "".join(chr(int(x)) for x in rng.integers(lo, hi, size=L))
        return seq, qual

    sam = ["@HD\tVN:1.6\tSO:unsorted", "@SQ\tSN:chr1\tLN:1000", "@PG\tID:x\tPN:y"]
    flag_choices = [0, 16, 65, 129, 81, 161, 97, 145, 256, 272, 512, 577, 1, 17, 0x41 | 0x200, 0x81 | 0x100]
    for i in range(300):
        L = int(rng.integers(1, 90))
        seq, qual = seq_qual(L)
        flag = int(rng.choice(flag_choices))
        if rng.random() < 0.1:
            qual = "*"
        extra = ["NM:i:0", "BX:Z:ACGT-1"][: int(rng.integers(0, 3))]
        sam.append("\t".join(["q%d" % i, str(flag), "chr1", "100", "60", "%dM" % L, "*", "0", "0", seq, qual] + extra))
    (tmp_path / "x.sam").write_text("\n".join(sam) + "\n")

    qseq, export = [], []
    for i in range(200):
        L = int(rng.integers(1, 80))
        seq, qual = seq_qual(L, 64, 105)
        seq = seq.replace("N", ".") if i % 7 == 0 else seq
        read = str(int(rng.choice([1, 2, 3])))
        index = str(int(rng.choice([0, 1, 7])))
        filt = str(int(rng.integers(0, 2)))
        qseq.append("\t".join(["M1", "42", "3", str(i % 9 + 1), str(1000 + i), str(2000 + i), index, read, seq, qual, filt]))
        export.append("\t".join(["M2", "7", "1", str(i % 5 + 1), str(10 + i), str(20 + i), index if index != "0" else "", read, seq,
                                 qual] + ["x"] * 11 + [["N", "Y"][int(filt)]]))
    (tmp_path / "x_qseq.txt").write_text("\n".join(qseq) + "\n")
    (tmp_path / "x_export.txt").write_text("\n".join(export) + "\n")
    return [str(tmp_path / "x.sam"), str(tmp_path / "x_qseq.txt"), str(tmp_path / "x_export.txt")]


@pytest.mark.skipif(not os.path.exists(REF_READER), reason="reference build not present")
def test_sam_qseq_export_match_reference_live(tmp_path):
    """SAM, qseq and export input (DataLayer/FastaReader.cpp:256-352)."""
    for path in _sam_and_qseq(tmp_path):
        for opts in OPTS + [["-q", "15", "-Q", "20", "--no-chastity"]]:
            ref = subprocess.run([REF_READER] + opts + [path], stdout=subprocess.PIPE, check=True).stdout
            assert ref.count(b"\n") > 50
            assert run_mine(opts, path) == ref, (path, opts)


@pytest.mark.parametrize("mapped", ["1", "0"])
@pytest.mark.parametrize("i", range(len(OPTS)))
def test_parallel_reader_matches_golden(i, mapped, monkeypatch):
    """SequenceReader: the file cut into blocks at record boundaries, each parsed by the same FastaReader
    logic on its own thread (-j 4), here with windows of 600 bytes so that records straddle windows --
    windows of the mapped file (the default) and windows filled by pread (ABG_READER_MMAP=0)."""
    monkeypatch.setenv("ABG_READER_MMAP", mapped)
    inp = os.path.join(GOLDEN, "reader_input.fq")
    want = open(os.path.join(GOLDEN, "reader_%d.tsv" % i), "rb").read()
    assert run_mine(OPTS[i] + ["-j", "4"], inp) == want
    monkeypatch.setenv("ABG_READER_WINDOW", "600")
    assert run_mine(OPTS[i] + ["-j", "3"], inp) == want
    monkeypatch.setenv("ABG_READER_WINDOW", "64")  # smaller than a record: the reader reads on
    assert run_mine(OPTS[i] + ["-j", "8"], inp) == want
    # the records wholesale, a parser thread's block at a time (what the host binary's PASS 1 takes)
    assert run_mine(OPTS[i] + ["-j", "8", "--blocks"], inp) == want
    monkeypatch.setenv("ABG_READER_WINDOW", "600")
    assert run_mine(OPTS[i] + ["-j", "3", "--blocks"], inp) == want
    assert run_mine(OPTS[i] + ["--blocks"], inp) == want  # (one thread: no block mode, the records one by one)


def test_parallel_reader_on_quality_lines_starting_with_at_and_plus(tmp_path, monkeypatch):
    """A quality line may begin with '@' or '+' (and a '+' line may repeat the id): the block boundaries must
    still fall on record starts.  Sequential reader == parallel reader for every window / thread count."""
    import numpy as np
    rng = np.random.default_rng(11)
    lines = []
    for i in range(3000):
        L = int(rng.integers(1, 160))
        seq = "".join(rng.choice(list("ACGTacgtN"), p=[.22, .22, .22, .22, .02, .02, .02, .02, .04], size=L))
        qual = "".join(chr(int(x)) for x in rng.integers(33, 74, size=L))
        r = rng.random()
        if r < 0.2:
            qual = "@" + qual[1:]
        elif r < 0.4:
            qual = "+" + qual[1:]
        casava = ["", " 1:N:0:ACGT", " 2:Y:0:ACGT"][int(rng.integers(0, 3))]
        plus = "+" if rng.random() < 0.7 else "+read%d" % i
        lines.append("@read%d%s\n%s\n%s\n%s\n" % (i, casava, seq, plus, qual))
    p = tmp_path / "x.fq"
    p.write_text("".join(lines))
    for opts in ([], ["-q", "3"], ["-q", "15", "-Q", "20", "--no-chastity"]):
        want = run_mine(opts, str(p))
        assert want.count(b"\n") > 1500
        for window, j in ((None, 2), (None, 7), ("5000", 4), ("100000", 16), ("300", 5)):
            if window:
                monkeypatch.setenv("ABG_READER_WINDOW", window)
            else:
                monkeypatch.delenv("ABG_READER_WINDOW", raising=False)
            assert run_mine(opts + ["-j", str(j)], str(p)) == want, (opts, window, j)


def test_compressed_input_inflated_ahead_yields_the_same_records(tmp_path):
    """abghost::Prefetch: a compressed input inflated into memory by a thread of its own and then parsed
    block-parallel (FASTQ) or sequentially (anything else) gives the records of the plain file."""
    import shutil
    for name, opts in (("reader_input.fq", ["-q", "3"]), ("reader_input.fq", []), ("reader_input.sam", [])):
        src = os.path.join(GOLDEN, name)
        if not os.path.exists(src):
            continue
        want = run_mine(opts, src)
        shutil.copy(src, tmp_path / name)
        subprocess.run(["gzip", "-kf", str(tmp_path / name)], check=True)
        gz = str(tmp_path / (name + ".gz"))
        assert run_mine(opts, gz) == want                       # the stream as it comes (one gunzip, sequential reader)
        assert run_mine(opts + ["-j", "4", "--prefetch"], gz) == want
        assert run_mine(opts + ["--prefetch"], gz) == want      # one thread: sequential reader over the inflated buffer


def _odd_fastq(rng, n, broken=None):
    """FASTQ text with everything the block parser (SequenceReader::parse_fastq_block) must either do exactly as
    FastaReader::read does or hand back to it: CR LF line ends, '#' lines, Casava comments with tabs and runs of
    blanks, ids that end in /1 already, lower-case ends, empty comments, quality below every threshold, a SAM-like
    header line; `broken`: one malformed record ("empty", "lengths", "noplus") in the middle."""
    out = []
    for i in range(n):
        L = int(rng.integers(1, 90))
        seq = "".join(rng.choice(list("ACGTacgtNn"), p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .02], size=L))
        lo = 33 if rng.random() < 0.8 else 34
        qual = "".join(chr(int(x)) for x in rng.integers(lo, [36, 74, 127][int(rng.integers(0, 3))], size=L))
        if qual[0] in "@+":
            qual = "I" + qual[1:]
        eol = "\r\n" if rng.random() < 0.3 else "\n"
        sep = [" ", "\t", "  \t "][int(rng.integers(0, 3))]
        casava = ["", sep + "1:N:0:ACGT", sep + "2:Y:0:ACGT", sep + "x", sep, sep + "1:N:", sep + "3:N:0"][int(rng.integers(0, 7))]
        name = ["read%d" % i, "r%d/1" % i, "x", "", "abc"][int(rng.integers(0, 5))]
        if rng.random() < 0.03:
            out.append("# a comment line" + eol)
        if rng.random() < 0.02:
            out.append("@CO\tsomething" + eol)
        if broken and i == n // 2:
            casava = ""
            if broken == "empty":
                seq, qual = "", ""
            elif broken == "lengths":
                qual = qual + "I"
            elif broken == "noplus":
                out.append("@%s%s%s%s%s-%s%s%s" % (name, casava, eol, seq, eol, eol, qual, eol))
                continue
        out.append("@%s%s%s%s%s+%s%s%s%s" % (name, casava, eol, seq, eol, "" if rng.random() < 0.7 else name, eol, qual, eol))
    return "".join(out)


@pytest.mark.parametrize("fast", ["1", "0"])
def test_block_parser_is_the_stream_reader_on_odd_input(tmp_path, monkeypatch, fast):
    """The records of the parser threads' blocks (ABG_READER_FAST=1: parsed where they lie; 0: every block through
    FastaReader over fmemopen) against the sequential FastaReader, record for record; a malformed record ends the run
    with the sequential reader's message, line number included."""
    import numpy as np
    build.build_hostcheck()
    monkeypatch.setenv("ABG_READER_FAST", fast)
    allopts = OPTS + [["-q", "15", "-Q", "20", "--no-chastity"], ["-Q", "3"], ["-q", "40"], ["--illumina-quality", "-q", "3"]]
    for seed in range(3):
        rng = np.random.default_rng(100 + seed)
        text = _odd_fastq(rng, 500)
        if seed == 1:
            text = text.rstrip("\r\n")            # no end of line after the last record
        if seed == 2:
            text = text[:text.rstrip("\r\n").rfind("\n") + 1]  # ... and no quality line at all
        p = tmp_path / ("odd%d.fq" % seed)
        p.write_bytes(text.encode())
        for opts in allopts:
            want = run_mine(opts, str(p))
            assert want.count(b"\n") > 150
            for window, j in ((None, 8), ("3000", 4), ("700", 3)):
                if window:
                    monkeypatch.setenv("ABG_READER_WINDOW", window)
                else:
                    monkeypatch.delenv("ABG_READER_WINDOW", raising=False)
                extra = ["--blocks"] if (j + len(opts)) % 2 else []
                assert run_mine(opts + ["-j", str(j)] + extra, str(p)) == want, (seed, opts, window, j, extra)
    monkeypatch.delenv("ABG_READER_WINDOW", raising=False)
    for broken in ("empty", "lengths", "noplus"):
        rng = np.random.default_rng(7)
        p = tmp_path / ("broken_%s.fq" % broken)
        p.write_bytes(_odd_fastq(rng, 300, broken).encode())
        seq = subprocess.run([build.READER_CHECK, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        par = subprocess.run([build.READER_CHECK, "-j", "4", "--blocks", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert seq.returncode != 0 and par.returncode == seq.returncode, broken
        assert par.stderr == seq.stderr and b"error" in seq.stderr, (broken, seq.stderr, par.stderr)
