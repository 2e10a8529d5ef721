"""pytest configuration: markers + shared fixtures.

``-m "not gpu"`` (runs anywhere): oracle vs golden vectors / compiled
reference, host logic, C-ABI symbol checks.  ``-m gpu``: parity tests proper,
through the C-ABI of libfmrx.so on a real MI355X.
"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (build container only)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from _oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libfmref.so not built (reference sources absent)")
    return Ref()


@pytest.fixture(scope="session")
def fmrx():
    """The product library through its Python host mirror (ctypes over the C-ABI).
    Builds it first (hipcc, gfx950) when the in-tree .so is missing, e.g. on a fresh checkout."""
    import importlib
    if not os.path.exists(os.path.join(ROOT, "software-defined-radio_amd", "lib", "libfmrx.so")):
        import __graft_entry__
        __graft_entry__.build()
    return importlib.import_module("software-defined-radio_amd")
