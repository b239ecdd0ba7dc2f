import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product library and the checkers once per session (no-ops when up to date).
    On the GPU box the prebuilt in-tree artefacts travel with the snapshot."""
    from abyss_amd import build
    if os.environ.get("ABG_TESTS_NO_BUILD"):  # (iterating on one piece while another is still compiling)
        yield
        return
    build.build_lib()
    build.build_oracle()
    build.build_hostcheck()
    # On a GPU box torch brings its own HIP runtime: it must open the device BEFORE libabyss_amd.so's
    # runtime does (the other order leaves torch without a GPU: "No HIP GPUs are available"), whatever
    # order the test files run in.  bench.py and __graft_entry__.smoke() have the same order.
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:  # noqa: BLE001
        pass
    yield
