import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently: only auto-skip when the
    # user did not ask for gpu tests explicitly.
    if _has_gpu():
        return
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _test_cus():
    """NNLM_TEST_CUS=n: the whole run plans its launches as if the device had n compute units (nnlm_debug_set_cus, a test hook of the C
    ABI) -- the deep fuzz runs of DESIGN.md section 2 put small random shapes through the persistent form of the SCD sweep this way.
    (Rounds 1-4 had the library read NNLM_DEBUG_CUS itself.)"""
    n = int(os.environ.get("NNLM_TEST_CUS", "0") or 0)
    if n > 0 and _has_gpu():
        from nnlm_amd import _lib
        _lib.debug_set_cus(n)
    yield
    if n > 0 and _has_gpu():
        from nnlm_amd import _lib
        _lib.debug_set_cus(0)
