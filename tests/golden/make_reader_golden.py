#!/usr/bin/env python3
"""Regenerates the sequence-reader fixtures: what the reference's own FastaReader (compiled
unmodified into oracle/_ref/ref_reader) returns for tests/golden/reader_input.* under each option
set of tests/test_host_reader.py.  Needs /root/reference (run `make -C oracle ref` first)."""
import os
import subprocess
import sys
import pathlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import test_host_reader as t  # noqa: E402


def main():
    assert os.path.exists(t.REF_READER), "build oracle/_ref first"
    inputs = {"": os.path.join(HERE, "reader_input.fq")}
    tmp = pathlib.Path(HERE)
    sam, qseq, export = t._sam_and_qseq(tmp)
    for src, name in ((sam, "reader_input.sam"), (qseq, "reader_input_qseq.txt"), (export, "reader_input_export.txt")):
        os.replace(src, os.path.join(HERE, name))
    inputs.update({"sam_": os.path.join(HERE, "reader_input.sam"), "qseq_": os.path.join(HERE, "reader_input_qseq.txt"),
                   "export_": os.path.join(HERE, "reader_input_export.txt")})
    for tag, path in inputs.items():
        for i, opts in enumerate(t.OPTS):
            out = subprocess.run([t.REF_READER] + opts + [path], stdout=subprocess.PIPE, check=True).stdout
            open(os.path.join(HERE, "reader_%s%d.tsv" % (tag, i)), "wb").write(out)
            print(os.path.basename(path), opts, out.count(b"\n"), "records")


if __name__ == "__main__":
    main()
