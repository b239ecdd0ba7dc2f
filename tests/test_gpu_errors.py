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


def test_a_sliced_context_waits_for_its_communicator():
    """abg_params.slice_filter = 1: the context holds no counters until a communicator says which range is its own; PASS 1
    before that is ABG_EINVAL with the way out in the message, and the pointer-level calls stay refused afterwards."""
    g = api.BloomDBG(32, counters=1 << 20, slice_filter=1)
    m1, m2 = synth.make_read_set(5000, 10.0)
    buf, off = api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))
    with pytest.raises(api.AbyssAmdError) as e:
        g.load(buf, off)
    assert "(%d)" % _lib.ABG_EINVAL in str(e.value) and "abg_attach_comm" in str(e.value), str(e.value)
    assert g.stats()["counter_bytes_held"] == 0
    from abyss_amd import dist as adist
    comm = adist.LocalComm()
    g.attach_comm(comm)
    assert g.stats()["counter_bytes_held"] == (1 << 20) + 64
    g.load(buf, off)
    _, contigs = g.assemble(buf, off)
    assert any(not c.redundant for c in contigs)
    with pytest.raises(api.AbyssAmdError) as e:
        g.contains_seq(bytes(buf[:int(off[1])]))
    assert "sliced" in str(e.value), str(e.value)
    g.close()


def test_a_filter_beyond_the_device_asks_for_ranks(monkeypatch):
    """slice_filter = 0 (the default): a filter that would not fit this device -- here a 64 MB one, ABG_MEM_LIMIT_MB -- is created
    without counters and says what it needs instead of running out of memory."""
    monkeypatch.setenv("ABG_MEM_LIMIT_MB", "64")
    g = api.BloomDBG(32, bloom_bytes=56 << 20)  # 49.8 M counters + 6.2 MB of visited bits
    assert g.stats()["counter_bytes_held"] == 0
    with pytest.raises(api.AbyssAmdError) as e:
        g.load(b"ACGT" * 20, np.array([0, 80], dtype=np.uint64))
    assert "(%d)" % _lib.ABG_EINVAL in str(e.value) and "does not fit the device" in str(e.value), str(e.value)
    g.close()
