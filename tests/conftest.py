import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_build():
    """Compile the CPU oracle (test infrastructure) into oracle/_build/."""
    subprocess.run(["make", "-s", "-f", "oracle/Makefile"], cwd=ROOT, check=True)
    return os.path.join(ROOT, "oracle", "_build")


@pytest.fixture(scope="session")
def oracle(oracle_build):
    import oracle_ctypes
    return oracle_ctypes.load(os.path.join(oracle_build, "libmia_oracle.so"))
