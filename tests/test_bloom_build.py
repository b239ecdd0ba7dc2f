"""`abyss-bloom build -t counting | -t rolling-hash -l N` (Bloom/bloom.cc:585-620): Bloom file
formats and the HashAgnosticCascadingBloom, against files written by the reference's own
classes (tests/golden/file_*.bloom, made with oracle/_ref/tier1)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from abyss_amd import api, build, synth
from util import GOLDEN

CASES = [("file_counting_k32_h3.bloom", "counting", 32, 3, 1, 262144),
         ("file_cascade_k32_h2_l2.bloom", "cascade", 32, 2, 2, 1 << 20),
         ("file_cascade_k24_h3_l3.bloom", "cascade", 24, 3, 3, 1 << 16)]


def golden_reads():
    m1, m2 = synth.make_read_set(20000, 30.0)
    return api.matrix_to_seqs(synth.codes_to_ascii(np.concatenate([m1, m2])))


def cascade_lib():
    l = ob.lib()
    l.orc_cascade_create.restype = C.c_void_p
    l.orc_cascade_create.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint64]
    l.orc_cascade_destroy.argtypes = [C.c_void_p]
    l.orc_cascade_load_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    l.orc_cascade_level.restype = C.POINTER(C.c_uint8)
    l.orc_cascade_level.argtypes = [C.c_void_p, C.c_uint]
    return l


def oracle_file(kind, k, H, levels, size, buf, off):
    if kind == "counting":
        o = ob.Oracle(k, counters=size, num_hashes=H, min_cov=0)
        o.load(buf, off)
        return api.counting_bloom_file(o.counters(), k, H)
    l = cascade_lib()
    h = l.orc_cascade_create(k, H, levels, size)
    l.orc_cascade_load_seqs(h, buf, off.ctypes.data, len(off) - 1)
    bits = np.ctypeslib.as_array(l.orc_cascade_level(h, levels - 1), (size // 8,)).copy()
    l.orc_cascade_destroy(h)
    return api.bit_bloom_file(bits, k, H)


@pytest.mark.parametrize("name,kind,k,H,levels,size", CASES)
def test_oracle_writes_the_reference_file(name, kind, k, H, levels, size):
    buf, off = golden_reads()
    assert oracle_file(kind, k, H, levels, size, buf, off) == open(os.path.join(GOLDEN, name), "rb").read()


@pytest.mark.parametrize("name,kind,k,H,levels,size", CASES[1:])
def test_device_logic_cascade(name, kind, k, H, levels, size):
    from test_hostcheck import HostCheck  # noqa: F401  (builds the library)
    l = C.CDLL(build.build_hostcheck())
    l.hc_create_cascade.restype = C.c_void_p
    l.hc_create_cascade.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint]
    l.hc_cascade_level.restype = C.POINTER(C.c_uint8)
    l.hc_cascade_level.argtypes = [C.c_void_p, C.c_uint]
    l.hc_load_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    l.hc_destroy.argtypes = [C.c_void_p]
    buf, off = golden_reads()
    h = l.hc_create_cascade(k, H, levels, size, 30000, 12)
    assert h
    assert l.hc_load_seqs(h, buf, off.ctypes.data, len(off) - 1) == 0
    bits = np.ctypeslib.as_array(l.hc_cascade_level(h, levels - 1), (size // 8,)).copy()
    l.hc_destroy(h)
    assert api.bit_bloom_file(bits, k, H) == open(os.path.join(GOLDEN, name), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,k,H,levels,size", CASES)
def test_gpu_builds_the_reference_file(name, kind, k, H, levels, size):
    buf, off = golden_reads()
    if kind == "counting":
        g = api.BloomDBG(k, counters=size, num_hashes=H, min_cov=0)
        g.load(buf, off)
        data = api.counting_bloom_file(g.counters(), k, H)
    else:
        g = api.BloomDBG(k, counters=size, num_hashes=H, min_cov=0, cascade_levels=levels)
        g.load(buf, off)
        data = api.bit_bloom_file(g.cascade_level(levels - 1), k, H)
    assert data == open(os.path.join(GOLDEN, name), "rb").read()


@pytest.mark.gpu
def test_gpu_abyss_bloom_cli_and_prebuilt_assembly(tmp_path):
    """abyss-bloom build -t counting | abyss-bloom-dbg -i (bloom-dbg.cc:302-343) == one-step assembly;
    abyss-bloom build -t rolling-hash -l 2 writes the reference's file."""
    cli = build.build_cli()
    bloom = os.path.join(os.path.dirname(cli), "abyss-bloom")
    m1, m2 = synth.make_read_set(20000, 30.0)
    reads = synth.codes_to_ascii(np.concatenate([m1, m2]))
    with open(tmp_path / "reads.fa", "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">r%d\n%s\n" % (i, bytes(s)))
    r = subprocess.run([bloom, "build", "-t", "counting", "-k32", "-H3", "-b262144", "c.bloom", "reads.fa"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert open(tmp_path / "c.bloom", "rb").read() == open(os.path.join(GOLDEN, "file_counting_k32_h3.bloom"), "rb").read()
    r = subprocess.run([bloom, "build", "-t", "rolling-hash", "-l2", "-k32", "-H2", "-b%d" % (2 * (1 << 20) // 8), "b.bloom", "reads.fa"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert open(tmp_path / "b.bloom", "rb").read() == open(os.path.join(GOLDEN, "file_cascade_k32_h2_l2.bloom"), "rb").read()
    # prebuilt filter path: same unitigs as building the filter in-process with the same number of counters
    one = subprocess.run([cli, "-k32", "-H3", "-b%d" % int(262144 * 1.125), "reads.fa"], cwd=tmp_path, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE)
    two = subprocess.run([cli, "-i", "c.bloom", "reads.fa"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr, two.stderr)
    assert one.stdout == two.stdout and len(two.stdout) > 1000
    if ob.have_ref():
        ref, _ = ob.run_ref(["-i", "c.bloom", "reads.fa"], cwd=str(tmp_path))
        assert ref == two.stdout
