import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference oracle/_ref/libref.so (development container)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    return binding.oracle()


@pytest.fixture(scope="session")
def ref_lib():
    from oracle import binding
    lib = binding.reference()
    if lib is None:
        pytest.skip("compiled reference (oracle/_ref/libref.so) not available here")
    return lib
