"""recometrics_amd -- MI355X-native drop-in for the ranking-metric hot path of david-cortes/recometrics.

Public surface = the reference's ``calc_reco_metrics`` (recometrics/__init__.py:44-628): same keyword arguments,
defaults, validation errors / warnings, dtype rule, output naming.  The per-user work (score the item pool, mask
train items, top-K, P/TP/R/AP/TAP/NDCG/Hit/RR@K, ROC/PR-AUC) runs in hand-written HIP kernels behind the C-ABI of
``include/recometrics_hip.h``.  There is no CPU fallback: without the built library and a GPU the call raises.
"""
import ctypes
import multiprocessing
import re
from warnings import warn

import numpy as np

from . import _binding

__all__ = ["calc_reco_metrics"]
__version__ = "0.1.0"

_FLAG_TO_KEY = (("precision", "p", "P@K"), ("trunc_precision", "tp", "TP@K"), ("recall", "r", "R@K"),
                ("average_precision", "ap", "AP@K"), ("trunc_average_precision", "tap", "TAP@K"),
                ("ndcg", "ndcg", "NDCG@K"), ("hit", "hit", "Hit@K"), ("rr", "rr", "RR@K"),
                ("roc_auc", "roc", "ROC_AUC"), ("pr_auc", "pr", "PR_AUC"))


def _row_major_with_ld(X):
    """(array, leading dimension in elements) -- reference __init__.py:11-16"""
    if X.flags["C_CONTIGUOUS"]:
        return X, X.shape[1]
    if X.strides[1] != X.dtype.itemsize:
        return np.ascontiguousarray(X), X.shape[1]
    return X, int(X.strides[0] / X.itemsize)


def _to_dtype(X, dtype):
    return X if X.dtype == dtype else X.astype(dtype)


def _sorted_csr_int32(X):
    """CSR with sorted indices and int32 index arrays -- reference __init__.py:26-41 (sorts in place, like it)"""
    from scipy.sparse import csr_array, issparse
    if issparse(X):
        if X.format != "csr":
            X = X.tocsr()
        X.sort_indices()
    else:
        X = csr_array(X)
    if X.indptr.dtype != np.int32 or X.indices.dtype != np.int32:
        X = X.copy()
        X.indptr = X.indptr.astype(np.int32)
        X.indices = X.indices.astype(np.int32)
    return X


def calc_reco_metrics(
    X_train, X_test,
    A, B,
    k=5,
    item_biases=None,
    as_df=True,
    precision=True,
    trunc_precision=False,
    recall=False,
    average_precision=True,
    trunc_average_precision=False,
    ndcg=True,
    hit=False,
    rr=False,
    roc_auc=False,
    pr_auc=False,
    all_metrics=False,
    rename_k=True,
    break_ties_with_noise=True,
    min_pos_test=1,
    min_items_pool=2,
    consider_cold_start=True,
    cumulative=False,
    nthreads=-1,
    seed=1,
):
    """Recommendation quality metrics for implicit-feedback models, evaluated per user on an MI355X.

    Arguments, defaults and return value follow ``recometrics.calc_reco_metrics`` (reference
    recometrics/__init__.py:44-413): ``X_train`` / ``X_test`` are CSR user-item matrices with the same shape,
    ``A`` [users, factors] and ``B`` [items, factors] are the model matrices (scores are ``A @ B.T``), ``k`` the
    cut-off.  Returns a ``pandas.DataFrame`` with one row per user (``as_df=True``) or a dict of arrays plus the
    entry ``"K"``.  Users that cannot be evaluated get NaN.

    Differences from the CPU reference, all documented in DESIGN.md: ``nthreads`` is accepted and ignored;
    ``break_ties_with_noise`` keeps its effect on the validity checks but exact score ties are broken by item id
    instead of by the reference's mt19937 noise stream; ``hit`` / ``rr`` requested alone are computed (the reference
    leaves them uninitialised) and ``pr_auc`` without ``roc_auc`` is computed from the full ranking.
    """
    import pandas as pd
    from scipy.sparse import csr_array, issparse

    if all_metrics:
        precision = trunc_precision = recall = average_precision = trunc_average_precision = True
        ndcg = hit = rr = roc_auc = pr_auc = True

    if item_biases is not None and isinstance(item_biases, pd.Series):
        item_biases = item_biases.to_numpy()

    if (A is None) != (B is None):
        raise ValueError("'A' and 'B' must either be passed together or passed as 'None' together.")
    if A is None:
        if item_biases is None:
            raise ValueError("Must pass item biases if not passing factors.")
        A = np.ones((X_test.shape[0], 1), dtype=ctypes.c_double, order="C")
        B = np.ascontiguousarray(item_biases, dtype=ctypes.c_double).reshape((-1, 1))
        item_biases = None

    assert isinstance(A, np.ndarray)
    assert isinstance(B, np.ndarray)
    assert issparse(X_test)

    if X_test.shape[0] >= np.iinfo(np.int32).max:
        raise ValueError("Number of test user is larger than maximum supported.")
    if X_test.shape[1] >= np.iinfo(np.int32).max:
        raise ValueError("Number of items is larger than maximum supported.")
    if not X_test.data.shape[0]:
        raise ValueError("'X_test' is empty.")
    if len(A.shape) != 2:
        raise ValueError("'A' must be a 2-dimensional array.")
    if len(B.shape) != 2:
        raise ValueError("'B' must be a 2-dimensional array.")
    if A.shape[1] != B.shape[1]:
        raise ValueError("'A' and 'B' must have the same number of columns.")
    if (not A.shape[0]) or (not A.shape[1]) or (not B.shape[1]) or (not X_test.shape[0]) or (not X_test.shape[1]):
        raise ValueError("Input matrices cannot be empty.")
    if A.shape[0] < X_test.shape[0]:
        raise ValueError("Number of users in 'A' and 'X_test' does not match.")
    if B.shape[0] < X_test.shape[1]:
        raise ValueError("Number of items in 'B' and 'X_test' does not match.")
    if A.shape[0] > X_test.shape[0]:
        warn("'A' has more users than 'X_test'.")
        A = A[:X_test.shape[0], :]
    if B.shape[0] > X_test.shape[1]:
        warn("'B' has more items than 'X_test'.")
        B = B[:X_test.shape[1], :]

    # float32 only when BOTH factor matrices are float32 (reference __init__.py:469)
    use_float = (A.dtype == ctypes.c_float) and (B.dtype == ctypes.c_float)
    dtype = np.float32 if use_float else np.float64

    if X_train is None:
        X_train = csr_array(X_test.shape, dtype=dtype)
        consider_cold_start = True
    assert issparse(X_train)
    assert X_train.shape[1] == X_test.shape[1]
    if X_train.shape[0] < X_test.shape[0]:
        raise ValueError("'X_train' and 'X_test' should have the same number of rows.")
    elif X_train.shape[0] > X_test.shape[0]:
        warn("'X_train' mas more rows than 'X_test'.")

    as_df, rename_k, cumulative = bool(as_df), bool(rename_k), bool(cumulative)
    break_ties_with_noise, consider_cold_start = bool(break_ties_with_noise), bool(consider_cold_start)
    flags = dict(precision=bool(precision), trunc_precision=bool(trunc_precision), recall=bool(recall),
                 average_precision=bool(average_precision), trunc_average_precision=bool(trunc_average_precision),
                 ndcg=bool(ndcg), hit=bool(hit), rr=bool(rr), roc_auc=bool(roc_auc), pr_auc=bool(pr_auc))
    if not (flags["precision"] or flags["average_precision"] or flags["ndcg"] or flags["hit"] or flags["rr"] or flags["roc_auc"]):
        raise ValueError("Must pass at least one metric to calculate.")

    if isinstance(seed, np.random.RandomState):
        seed = int(seed.randint(np.iinfo(np.int32).max))
    elif isinstance(seed, np.random.Generator):
        seed = int(seed.integers(np.iinfo(np.int32).max))
    nthreads, seed, k = int(nthreads), int(seed), int(k)
    min_pos_test, min_items_pool = int(min_pos_test), int(min_items_pool)
    assert seed >= 1
    assert k >= 1
    assert min_pos_test >= 1
    assert min_items_pool >= 1
    if nthreads < 0:
        nthreads = multiprocessing.cpu_count() + 1 + nthreads
    assert nthreads > 0
    if nthreads > 1 and not _binding.has_openmp():
        warn("Attempting to use more than 1 thread, but package was built without multi-threading support.")

    if k > X_test.shape[1]:
        raise ValueError("'k' should be smaller than the number of items.")

    if item_biases is not None:
        assert isinstance(item_biases, np.ndarray)
        if len(item_biases.shape) > 2:
            raise ValueError("'item_biases' should be a 1-d array.")
        if len(item_biases.shape) != 1:
            item_biases = item_biases.reshape(-1)
        if not item_biases.shape[0]:
            raise ValueError("'item_biases' is empty.")
        if item_biases.shape[0] < X_test.shape[1]:
            raise ValueError("Number of items in 'item_biases' must match with 'X_test'.")
        if item_biases.shape[0] > X_test.shape[1]:
            item_biases = item_biases[:X_test.shape[1]]
            warn("'item_biases' has more items than 'X_test'.")
        # fold the biases in as one more factor: A = [A | 1], B = [B | bias]   (reference __init__.py:549-550)
        item_biases = _to_dtype(item_biases, dtype)
        A = np.c_[A, np.ones((A.shape[0], 1), dtype=dtype)]
        B = np.c_[B, item_biases.reshape((-1, 1))]

    X_train = _sorted_csr_int32(X_train)
    X_test = _sorted_csr_int32(X_test)
    if X_test.dtype != dtype:
        X_test = X_test.astype(dtype)
    if X_train.shape[0] > X_test.shape[0]:
        X_train = X_train[:X_test.shape[0], :]
        X_train = _sorted_csr_int32(X_train)
    A, lda = _row_major_with_ld(_to_dtype(A, dtype))
    B, ldb = _row_major_with_ld(_to_dtype(B, dtype))

    want = {short: flags[flag] for flag, short, _ in _FLAG_TO_KEY}
    arrays = _binding.calc_metrics(
        A, lda, B, ldb, X_train.indptr, X_train.indices, X_test.indptr, X_test.indices, X_test.data,
        k, want, cumulative, break_ties_with_noise, consider_cold_start, min_items_pool, min_pos_test, nthreads, seed)

    out = {}
    for (flag, short, key), arr in zip(_FLAG_TO_KEY, arrays):
        if arr.shape[0]:
            out[key] = arr
    if not as_df:
        out["K"] = k
        return out
    if not cumulative:
        out = pd.DataFrame(out)
        if rename_k:
            out.columns = out.columns.str.replace("@K$", "@" + str(k), regex=True)
        return out
    frames = []
    for key, v in out.items():
        if v.ndim == 1:                      # ROC_AUC / PR_AUC stay single columns (the reference raises IndexError here)
            frames.append(pd.DataFrame({key: v}))
        else:
            frames.append(pd.DataFrame(v, columns=[re.sub("(@)K$", r"\1", key) + str(i + 1) for i in range(v.shape[1])]))
    return pd.concat(frames, axis=1)
