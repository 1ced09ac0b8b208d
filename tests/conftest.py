"""pytest configuration: markers, paths, and on-demand build of the TEST-ONLY oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE = os.path.join(ROOT, "oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_build():
    """Builds oracle/_build (C restatement; gcc only, no reference needed) and returns its directory."""
    subprocess.run(["make", "-C", ORACLE, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE, "_build")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
