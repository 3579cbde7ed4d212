"""CPU-only: the C-ABI library is built, loads, and exports every symbol include/recometrics_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "recometrics_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_boundary():
    syms = _declared_symbols()
    for must in ("rm_calc_metrics_f32", "rm_calc_metrics_f64", "rm_has_openmp", "rm_last_error"):
        assert must in syms


def test_library_builds_and_exports_every_declared_symbol():
    from recometrics_amd import build as rb
    path = rb.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), "missing export: " + sym
    from recometrics_amd import _binding
    assert set(_binding.EXPORTS) == set(_declared_symbols())
    # host-only calls that need no GPU
    lib.rm_last_error.restype = ctypes.c_char_p
    assert lib.rm_has_openmp() == 1
    assert lib.rm_device_count() >= 0
    assert lib.rm_last_error() is not None


def test_the_library_is_built_from_the_sources_as_they_are():
    """build() decides on content, and records what it compiled: the digest in csrc/build_stamp.json must be the digest of the
    sources in the tree (a stale .so next to newer sources -- or next to a reverted file with an old mtime -- is rebuilt)"""
    from recometrics_amd import build as rb
    rb.build()
    st = rb.read_stamp()
    assert st is not None and st["sources_sha256"] == rb.sources_digest()
    assert st["library_bytes"] == os.path.getsize(rb.LIB) and not rb.needs_build()
    assert set(st["translation_units"]) == set(rb.SOURCES)


def test_no_device_is_a_loud_error_not_a_fallback():
    """Without a GPU the product path must raise (status != 0, message set), never compute on the CPU."""
    import numpy as np
    from recometrics_amd import _binding
    if _binding.device_count() > 0:
        pytest.skip("a GPU is present")
    A = np.ones((2, 4), np.float32)
    B = np.ones((8, 4), np.float32)
    p = np.array([0, 1, 2], np.int32)
    i = np.array([1, 2], np.int32)
    with pytest.raises((RuntimeError, MemoryError)):
        _binding.rank(A, B, np.zeros(3, np.int32), np.zeros(0, np.int32), p, i, 2)


def test_product_never_imports_the_oracle():
    """recometrics_amd/ must not import, include, link or dlopen anything under oracle/."""
    pkg = os.path.join(ROOT, "recometrics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".h")):
                continue
            for line in open(os.path.join(dirpath, f), errors="replace"):
                low = line.lower()
                if "oracle" in low and ("import" in low or "#include" in low or "cdll" in low or "dlopen" in low):
                    raise AssertionError("%s references the oracle: %s" % (f, line.strip()))


def test_cython_binding_builds_and_imports():
    """recometrics_amd/_cy.pyx compiles against include/recometrics_hip.h and links to the library (no GPU needed)"""
    from recometrics_amd import build as rb
    path = rb.build_cython()
    assert os.path.exists(path)
    from recometrics_amd import _cy
    assert _cy.has_openmp() is True
    assert _cy.device_count() >= 0
    assert _cy.METRIC_ORDER == ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")
