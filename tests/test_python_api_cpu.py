"""CPU-only: the host-side mirror of recometrics.calc_reco_metrics (reference recometrics/__init__.py:414-562) --
argument validation, warnings and CSR normalisation happen before any device work, so they are testable without a GPU."""
import warnings

import numpy as np
import pytest
from scipy.sparse import csr_array

from recometrics_amd import calc_reco_metrics
from recometrics_amd import __init__ as _api  # noqa: F401


def _inputs(m=6, n=9, k=3, dtype=np.float32):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, k)).astype(dtype)
    B = rng.standard_normal((n, k)).astype(dtype)
    X = (rng.random((m, n)) < 0.3).astype(dtype)
    X[:, 0] = 1
    return csr_array(X * 0), csr_array(X), A, B


def test_validation_errors_match_the_reference_conditions():
    Xtr, Xte, A, B = _inputs()
    with pytest.raises(ValueError, match="must either be passed together"):
        calc_reco_metrics(Xtr, Xte, A, None)
    with pytest.raises(ValueError, match="Must pass item biases"):
        calc_reco_metrics(Xtr, Xte, None, None)
    with pytest.raises(ValueError, match="same number of columns"):
        calc_reco_metrics(Xtr, Xte, A, B[:, :2])
    with pytest.raises(ValueError, match="2-dimensional"):
        calc_reco_metrics(Xtr, Xte, A[0], B)
    with pytest.raises(ValueError, match="Number of users"):
        calc_reco_metrics(Xtr, Xte, A[:3], B)
    with pytest.raises(ValueError, match="Number of items"):
        calc_reco_metrics(Xtr, Xte, A, B[:4])
    with pytest.raises(ValueError, match="at least one metric"):
        calc_reco_metrics(Xtr, Xte, A, B, precision=False, average_precision=False, ndcg=False)
    with pytest.raises(ValueError, match="'k' should be smaller"):
        calc_reco_metrics(Xtr, Xte, A, B, k=100)
    with pytest.raises(ValueError, match="'X_test' is empty"):
        calc_reco_metrics(Xtr, csr_array((6, 9), dtype=np.float32), A, B)
    with pytest.raises(ValueError, match="same number of rows"):
        calc_reco_metrics(Xtr[:3], Xte, A, B)
    with pytest.raises(AssertionError):
        calc_reco_metrics(Xtr, Xte, A, B, k=0)
    with pytest.raises(ValueError, match="1-d array"):
        calc_reco_metrics(Xtr, Xte, A, B, item_biases=np.zeros((9, 1, 1), np.float32))


def test_helpers_follow_the_reference_rules():
    import recometrics_amd as ra
    X = np.arange(12, dtype=np.float32).reshape(3, 4)
    a, ld = ra._row_major_with_ld(X)
    assert a is X and ld == 4
    sub = np.arange(24, dtype=np.float32).reshape(3, 8)[:, :4]          # row-major view with a larger leading dimension
    a, ld = ra._row_major_with_ld(sub)
    assert a is sub and ld == 8
    a, ld = ra._row_major_with_ld(np.asfortranarray(X))                 # column-major => copied
    assert a.flags["C_CONTIGUOUS"] and ld == 4
    Xs = csr_array(np.array([[0, 2, 1], [3, 0, 0]], dtype=np.float64))
    Xs.indices = Xs.indices[::-1].copy() if False else Xs.indices
    out = ra._sorted_csr_int32(Xs)
    assert out.indptr.dtype == np.int32 and out.indices.dtype == np.int32 and out.has_sorted_indices


def test_missing_device_raises_instead_of_falling_back():
    from recometrics_amd import _binding
    if _binding.device_count() > 0:
        pytest.skip("a GPU is present")
    Xtr, Xte, A, B = _inputs()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises((RuntimeError, MemoryError)):
            calc_reco_metrics(Xtr, Xte, A, B, k=2)
