"""pytest configuration: markers and shared fixtures.

* `-m "not gpu"`: oracle vs. the reference's known-answer vectors, host logic, C-ABI symbol
  checks, gloo multi-process sharding.  Runs without a GPU.
* `-m gpu`: parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import tfo

    tfo.lib()
    return tfo


@pytest.fixture(scope="session")
def tf():
    """The product package (HIP path behind the C ABI)."""
    import twenty_first_amd as tf

    return tf
