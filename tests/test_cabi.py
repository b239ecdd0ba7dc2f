"""The C-ABI shared library: loads, exports every symbol of include/abyss_amd.h, and
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from abyss_amd import _lib, api, build


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(build.ROOT, "include", "abyss_amd.h")).read()
    declared = set(re.findall(r"\b(abg_[a-z_]+)\s*\(", header))
    declared -= {"abg_contig_cb"}
    assert declared == set(_lib.symbols())
    lib = C.CDLL(build.build_lib())
    for s in declared:
        assert hasattr(lib, s), s


def test_params_init_defaults():
    lib = _lib.load()
    p = _lib.Params()
    lib.abg_params_init(C.byref(p))
    # AssemblyParams defaults, AssemblyParams.h:78-85
    assert (p.num_hashes, p.min_cov, p.trim) == (4, 2, 0xFFFFFFFF)


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="a GPU is present")
def test_no_cpu_fallback():
    with pytest.raises(api.AbyssAmdError) as e:
        api.BloomDBG(32, counters=4096)
    assert "no HIP device" in str(e.value)


def test_rejects_bad_parameters():
    lib = _lib.load()
    for k, h in ((1, 4), (193, 4), (32, 0), (32, 33)):
        p = _lib.Params()
        lib.abg_params_init(C.byref(p))
        p.k, p.num_hashes, p.counters = k, h, 4096
        ctx = C.c_void_p()
        assert lib.abg_create(C.byref(p), C.byref(ctx)) == -1
        assert lib.abg_last_error(None)


def test_host_binary_multi_gpu_launch_fails_cleanly_without_devices(tmp_path):
    """`abyss-bloom-dbg --gpus 2` forks one process per GPU before touching the HIP runtime; where there
    is no GPU (this container) every rank must give up with a message and exit 1 -- nobody waits for ever."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    from abyss_amd import build
    exe = build.build_cli()
    (tmp_path / "r.fa").write_text(">a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([exe, "-k32", "-b1M", "--gpus=2", "r.fa"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and r.stdout == b""
    r = subprocess.run([exe, "-k32", "-b1M", "--gpus=2", "--checkpoint=100", "r.fa"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1 and b"--checkpoint is not available with --gpus" in r.stderr


@pytest.mark.skipif(_have_gpu(), reason="a GPU is present")
def test_rresolver_has_no_cpu_fallback(tmp_path):
    """abg_rr_create -> ABG_ENODEV without a GPU, and the drop-in abyss-rresolver-short says so and exits 1 once it needs the filter."""
    import subprocess
    import rr_util
    with pytest.raises(api.AbyssAmdError) as e:
        api.ReadFilter(1 << 20, 64)
    assert "no HIP device" in str(e.value)
    build.build_cli()
    reads = rr_util.write_inputs(str(tmp_path), "rr_k32")
    r = subprocess.run([os.path.join(build.BIN_DIR, "abyss-rresolver-short"), "-b8M", "-k32", "-c", "o.fa", "-g", "o.dot", "--dot", "rr_k32-1.fa", "rr_k32-1.dot"] + reads,
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1 and b"no HIP device" in r.stderr and not os.path.exists(tmp_path / "o.fa")
