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
    from oracle.oracle import Oracle
    return Oracle()


def _reload_library_switches():
    """The library reads its RM_DEBUG_* / RM_*_MB switches once, at load (csrc/rm_lib.hip `Switches`); tests that change the
    environment ask it to read them again (rm_debug_reload_switches).  Only when the library is already loaded: CPU-only tests
    that never touch it must not trigger a build."""
    mod = sys.modules.get("recometrics_amd._binding")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        mod._lib.rm_debug_reload_switches()


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment(monkeypatch):
    """monkeypatch.setenv / delenv inside a test take effect in the library at once; at the start of every test the library sees
    the environment as the previous test's teardown left it."""
    _reload_library_switches()
    set_orig, del_orig = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_orig(name, value, *a, **k)
        if name.startswith("RM_"):
            _reload_library_switches()

    def delenv(name, *a, **k):
        del_orig(name, *a, **k)
        if name.startswith("RM_"):
            _reload_library_switches()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
