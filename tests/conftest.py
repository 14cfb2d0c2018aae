import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref, i.e. the build container with /root/reference")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle, build_oracle
    build_oracle()
    return Oracle()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def qpsk_lib():
    """The built C-ABI library (CPU tests only check that it loads and what it exports)."""
    import qpsk_amd
    if not os.path.exists(qpsk_amd.lib_path()):
        qpsk_amd.build()
    return qpsk_amd.load()
