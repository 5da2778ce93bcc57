import os
import sys

import pytest

# the kernel-form hook nnop_debug_set (csrc/nnop_debug.h) is locked unless the process starts with this (read at the library's first launch)
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def tune(pkg):
    """Override launch-shape knobs of the library for one test (csrc/nnop_debug.h) and restore them afterwards.
    `tune(fwd_split=0, fwd_nw=4)`; -1 = automatic."""
    saved = {}

    def _set(**kw):
        for k, v in kw.items():
            prev = pkg._lib.debug_set(k, v)
            saved.setdefault(k, prev)

    yield _set
    for k, v in saved.items():
        pkg._lib.debug_set(k, v)
