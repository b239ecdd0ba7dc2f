"""The C ABI's promise on failure (include/abyss_amd.h): a negative ABG_E* code and a message from
abg_last_error(), never an exit of the host process.  ABG_MEM_LIMIT_MB gives a context a device
memory budget, so "the device is too small" can be tested on a 288 GB MI355X: abg_create fails
cleanly when the filters do not fit, abg_load_seqs when PASS 1's scratch does not, and the drop-in
binary turns the library's error into the reference's way of failing -- a message and exit status 1
(Common/IOUtil.h:14-22), not an abort()."""
import subprocess

import numpy as np
import pytest

from abyss_amd import api, build, synth, _lib

pytestmark = pytest.mark.gpu


def test_create_fails_cleanly_when_the_filters_do_not_fit(monkeypatch):
    monkeypatch.setenv("ABG_MEM_LIMIT_MB", "64")
    with pytest.raises(api.AbyssAmdError) as e:
        api.BloomDBG(64, bloom_bytes=1 << 30)  # 0.95 GB of counters + 0.12 GB of visited bits
    assert "(%d)" % _lib.ABG_ENOMEM in str(e.value) and "no device memory" in str(e.value), str(e.value)
    # the process goes on, and a context that fits still works
    monkeypatch.delenv("ABG_MEM_LIMIT_MB")
    m1, m2 = synth.make_read_set(20000, 20.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    g = api.BloomDBG(64, bloom_bytes=16 << 20)
    g.load(buf, off)
    _, contigs = g.assemble(buf, off)
    assert any(not c.redundant for c in contigs)
    g.close()


def test_load_fails_cleanly_when_pass1_scratch_does_not_fit(monkeypatch):
    # the filters fit the budget (B=256M: 0.27 GB with the visited bits), PASS 1's hashes, bins and claim
    # tables (several times that) do not
    monkeypatch.setenv("ABG_MEM_LIMIT_MB", "400")
    g = api.BloomDBG(64, bloom_bytes=256 << 20)
    m1, m2 = synth.make_read_set(200000, 30.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    with pytest.raises(api.AbyssAmdError) as e:
        g.load(buf, off)
    assert "(%d)" % _lib.ABG_ENOMEM in str(e.value), str(e.value)
    g.close()  # (all such a context is still good for)


def test_binary_reports_the_error_and_exits_with_status_1(tmp_path, monkeypatch):
    m1, m2 = synth.make_read_set(50000, 20.0)
    synth.write_fastq(str(tmp_path / "r1.fq"), m1, "r", 1)
    synth.write_fastq(str(tmp_path / "r2.fq"), m2, "r", 2)
    monkeypatch.setenv("ABG_MEM_LIMIT_MB", "64")
    r = subprocess.run([build.build_cli(), "-k64", "-b1G", "-j4", "r1.fq", "r2.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"no device memory" in r.stderr and r.stdout == b"", (r.returncode, r.stderr[-500:])
