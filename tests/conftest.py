import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces built once per session (product library, oracle, test-only host walker)."""
    import subprocess
    import forgex_amd
    if not os.path.exists(forgex_amd.LIB_PATH):
        forgex_amd.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "support")])
    return True
