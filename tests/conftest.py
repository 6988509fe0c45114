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


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with one addition: the library reads its FXAMD_* hooks from the environment once per process
    (fx_env, csrc/fxamd.hip), so setting or deleting one of them -- and undoing that at the end of the test -- makes it read
    them again (fxamd_reload_env of include/forgex_amd_bench.h)."""
    import forgex_amd._lib as _lib

    def reload_env():
        if _lib._lib is not None:
            _lib._lib.fxamd_reload_env()

    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        if name.startswith("FXAMD_"):
            reload_env()

    def delenv(name, raising=True):
        orig_del(name, raising)
        if name.startswith("FXAMD_"):
            reload_env()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield monkeypatch
    monkeypatch.undo()
    reload_env()
