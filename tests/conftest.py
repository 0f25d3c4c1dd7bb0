import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is test infrastructure: build it on demand (gcc only, no GPU needed)
    lib = os.path.join(ROOT, "oracle", "libkosk_oracle.so")
    src = os.path.join(ROOT, "oracle", "kosk_oracle.c")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "--no-print-directory"])


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib
