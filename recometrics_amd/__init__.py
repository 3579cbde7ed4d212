"""recometrics_amd -- MI355X-native drop-in for the ranking-metric hot path of david-cortes/recometrics.

Public surface = the reference's ``calc_reco_metrics`` (recometrics/__init__.py:44-628): same keyword arguments,
defaults, error / warning conditions, dtype rule and output naming.  The per-user work (score the item pool, mask
train items, top-K, P/TP/R/AP/TAP/NDCG/Hit/RR@K, ROC/PR-AUC) runs in hand-written HIP kernels behind the C-ABI of
``include/recometrics_hip.h``.  There is no CPU fallback: without the built library and a GPU the call raises.
"""
import multiprocessing
from warnings import warn

import numpy as np

from . import _binding

__all__ = ["calc_reco_metrics", "split_reco_train_test"]
__version__ = "0.1.0"

# (keyword of calc_reco_metrics, name in the C-ABI order, key of the result dict) -- reference __init__.py:590-613
_METRICS = (
    ("precision", "p", "P@K"), ("trunc_precision", "tp", "TP@K"), ("recall", "r", "R@K"),
    ("average_precision", "ap", "AP@K"), ("trunc_average_precision", "tap", "TAP@K"), ("ndcg", "ndcg", "NDCG@K"),
    ("hit", "hit", "Hit@K"), ("rr", "rr", "RR@K"), ("roc_auc", "roc", "ROC_AUC"), ("pr_auc", "pr", "PR_AUC"),
)
_INT32_MAX = np.iinfo(np.int32).max


def split_reco_train_test(*args, **kwargs):
    """See :func:`recometrics_amd.split.split_reco_train_test` (imported lazily: it needs SciPy)."""
    from .split import split_reco_train_test as impl
    return impl(*args, **kwargs)


def _row_major_with_ld(X):
    """(array, leading dimension in elements): a row-major view keeps its stride, anything else is copied
    (reference __init__.py:11-16)."""
    if X.flags["C_CONTIGUOUS"]:
        return X, X.shape[1]
    st0 = X.strides[0]
    if X.strides[1] == X.dtype.itemsize and st0 > 0 and st0 % X.itemsize == 0 and st0 // X.itemsize >= X.shape[1]:
        return X, st0 // X.itemsize
    return np.ascontiguousarray(X), X.shape[1]          # reversed / overlapping / unaligned rows: a size_t cannot express them


def _csr_int32(X):
    """CSR with int32 index arrays over contiguous buffers (reference __init__.py:26-41), rows AS THEY ARE.

    The reference sorts the column indices of every row with SciPy's `sort_indices()` -- one thread walking every stored entry
    whenever SciPy does not already know the answer, 26-33 ms for the two matrices of BASELINE C2 in front of a 10 ms device
    call.  Here the rows go to the library as they come: it validates them on the device (sorted? every index in range? index
    pointers monotone?  csrc/rm_prep.hpp k_check_csr_*, ~40 us) and only when some row is not sorted does it sort a COPY on
    the host (csrc/rm_csr.cpp) and run again.  Unlike the reference, the caller's matrix is never modified."""
    from scipy.sparse import csr_array, issparse
    X = X.tocsr() if issparse(X) and X.format != "csr" else (X if issparse(X) else csr_array(X))
    if X.indptr.dtype != np.int32 or X.indices.dtype != np.int32:
        X = csr_array((X.data, X.indices.astype(np.int32), X.indptr.astype(np.int32)), shape=X.shape, copy=False)
    return X


def _sorted_csr_int32(X, nthreads=0):
    """CSR with sorted column indices and int32 index arrays (reference __init__.py:26-41), for the HOST-side split
    (split.py: its C++ walks the rows on the CPU and must see what the reference sees).

    A matrix whose index arrays are int32 already is sorted in place, like the reference does; one with int64 indices is
    copied first (the reference sorts the caller's matrix, then copies).  SciPy's cached answer (`_has_sorted_indices`, the
    attribute behind the public `has_sorted_indices` property, whose getter runs SciPy's single-threaded check when the answer
    is unknown) is trusted when it says True; tests/test_csr_cpu.py pins the attribute's name for the installed SciPy --
    should a release rename it, every call merely pays for the library's check again.  Otherwise the check -- and the sort,
    should it be needed -- run in the library on `nthreads` host threads (csrc/rm_csr.cpp) and the flag is set."""
    from scipy.sparse import csr_array, issparse
    X = X.tocsr() if issparse(X) and X.format != "csr" else (X if issparse(X) else csr_array(X))
    if X.indptr.dtype != np.int32 or X.indices.dtype != np.int32:
        known = getattr(X, "_has_sorted_indices", None)
        X = X.copy()
        X.indptr, X.indices = X.indptr.astype(np.int32), X.indices.astype(np.int32)
        if known:
            X.has_sorted_indices = True
    if getattr(X, "_has_sorted_indices", None) is True:       # SciPy's cached answer (set by its own sort / check, or by us below)
        return X
    if not (X.indices.flags.c_contiguous and X.indptr.flags.c_contiguous and X.data.flags.c_contiguous
            and X.data.dtype.itemsize in (4, 8) and X.indices.flags.writeable and X.data.flags.writeable):
        X.sort_indices()                                        # unusual storage: SciPy's own pass
        return X
    if not _binding.csr_rows_sorted(X.indptr, X.indices, nthreads):
        _binding.csr_sort_rows(X.indptr, X.indices, X.data, nthreads)
    X.has_sorted_indices = True
    return X


def _fail_if(cond, message):
    if cond:
        raise ValueError(message)


def _check_factors(A, B, n_users, n_items):
    """Shape rules of reference __init__.py:442-467; returns A, B trimmed to the test matrix (with its warnings)."""
    _fail_if(A.ndim != 2, "'A' must be a 2-dimensional array.")
    _fail_if(B.ndim != 2, "'B' must be a 2-dimensional array.")
    _fail_if(A.shape[1] != B.shape[1], "'A' and 'B' must have the same number of columns.")
    _fail_if(0 in (A.shape[0], A.shape[1], B.shape[1], n_users, n_items), "Input matrices cannot be empty.")
    _fail_if(A.shape[0] < n_users, "Number of users in 'A' and 'X_test' does not match.")
    _fail_if(B.shape[0] < n_items, "Number of items in 'B' and 'X_test' does not match.")
    if A.shape[0] > n_users:
        warn("'A' has more users than 'X_test'.")
        A = A[:n_users]
    if B.shape[0] > n_items:
        warn("'B' has more items than 'X_test'.")
        B = B[:n_items]
    return A, B


def _fold_item_biases(A, B, item_biases, n_items, dtype):
    """scores = A @ B.T + bias  ==  [A | 1] @ [B | bias].T   (reference __init__.py:534-551)"""
    assert isinstance(item_biases, np.ndarray)
    _fail_if(item_biases.ndim > 2, "'item_biases' should be a 1-d array.")
    item_biases = item_biases.reshape(-1)
    _fail_if(item_biases.shape[0] == 0, "'item_biases' is empty.")
    _fail_if(item_biases.shape[0] < n_items, "Number of items in 'item_biases' must match with 'X_test'.")
    if item_biases.shape[0] > n_items:
        warn("'item_biases' has more items than 'X_test'.")
        item_biases = item_biases[:n_items]
    ones = np.ones((A.shape[0], 1), dtype=dtype)
    return np.hstack([A.astype(dtype, copy=False), ones]), np.hstack([B.astype(dtype, copy=False), item_biases.astype(dtype).reshape(-1, 1)])


def _as_frame(out, k, cumulative, rename_k):
    """dict of arrays -> DataFrame with the reference's column names (__init__.py:615-626): "P@K" -> "P@5", cumulative
    blocks -> "P@1" .. "P@k".  ROC_AUC / PR_AUC stay single columns in cumulative mode (the reference raises there)."""
    import pandas as pd
    if not cumulative:
        df = pd.DataFrame(out)
        if rename_k:
            df.columns = [c[:-1] + str(k) if c.endswith("@K") else c for c in df.columns]
        return df
    blocks = []
    for key, arr in out.items():
        if arr.ndim == 1:
            blocks.append(pd.DataFrame({key: arr}))
        else:
            blocks.append(pd.DataFrame(arr, columns=["%s%d" % (key[:-1], i + 1) for i in range(arr.shape[1])]))
    return pd.concat(blocks, axis=1)


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

    Differences from the CPU reference, all documented in DESIGN.md: ``nthreads`` only drives the host-side sort of a matrix
    whose rows turn out unsorted (the reference sorts ``X_train`` / ``X_test`` in place; here they are never modified);
    ``break_ties_with_noise`` adds the reference's own noise (``std::mt19937(seed + user)``, reproduced bit for bit on
    the device), scores that are still exactly equal afterwards are ordered by item id; ``hit`` / ``rr`` requested
    alone are computed (the reference leaves them uninitialised) and ``pr_auc`` without ``roc_auc`` is computed from
    the full ranking.
    """
    from scipy.sparse import csr_array, issparse

    requested = dict(precision=precision, trunc_precision=trunc_precision, recall=recall,
                     average_precision=average_precision, trunc_average_precision=trunc_average_precision,
                     ndcg=ndcg, hit=hit, rr=rr, roc_auc=roc_auc, pr_auc=pr_auc)
    requested = {name: bool(all_metrics or flag) for name, flag in requested.items()}

    if hasattr(item_biases, "to_numpy"):                 # pandas.Series
        item_biases = item_biases.to_numpy()

    # ---- factors (or the non-personalised mode: A = ones, B = biases; reference :429-436) ----
    _fail_if((A is None) != (B is None), "'A' and 'B' must either be passed together or passed as 'None' together.")
    if A is None:
        _fail_if(item_biases is None, "Must pass item biases if not passing factors.")
        A = np.ones((X_test.shape[0], 1), dtype=np.float64)
        B = np.ascontiguousarray(item_biases, dtype=np.float64).reshape(-1, 1)
        item_biases = None
    assert isinstance(A, np.ndarray) and isinstance(B, np.ndarray)
    assert issparse(X_test)
    n_users, n_items = X_test.shape
    _fail_if(n_users >= _INT32_MAX, "Number of test user is larger than maximum supported.")
    _fail_if(n_items >= _INT32_MAX, "Number of items is larger than maximum supported.")
    _fail_if(X_test.data.shape[0] == 0, "'X_test' is empty.")
    A, B = _check_factors(A, B, n_users, n_items)

    # float32 only when BOTH factor matrices are float32 (reference :469)
    dtype = np.float32 if (A.dtype == np.float32 and B.dtype == np.float32) else np.float64

    # ---- train matrix ----
    if X_train is None:
        X_train = csr_array(X_test.shape, dtype=dtype)
        consider_cold_start = True
    assert issparse(X_train)
    assert X_train.shape[1] == n_items
    _fail_if(X_train.shape[0] < n_users, "'X_train' and 'X_test' should have the same number of rows.")
    if X_train.shape[0] > n_users:
        warn("'X_train' mas more rows than 'X_test'.")

    _fail_if(not (requested["precision"] or requested["average_precision"] or requested["ndcg"]
                  or requested["hit"] or requested["rr"] or requested["roc_auc"]),
             "Must pass at least one metric to calculate.")

    # ---- scalars ----
    if isinstance(seed, np.random.RandomState):
        seed = seed.randint(_INT32_MAX)
    elif isinstance(seed, np.random.Generator):
        seed = seed.integers(_INT32_MAX)
    seed, k, nthreads = int(seed), int(k), int(nthreads)
    min_pos_test, min_items_pool = int(min_pos_test), int(min_items_pool)
    assert seed >= 1 and k >= 1 and min_pos_test >= 1 and min_items_pool >= 1
    if nthreads < 0:
        nthreads += multiprocessing.cpu_count() + 1
    assert nthreads > 0
    if nthreads > 1 and not _binding.has_openmp():
        warn("Attempting to use more than 1 thread, but package was built without multi-threading support.")
    _fail_if(k > n_items, "'k' should be smaller than the number of items.")

    if item_biases is not None:
        A, B = _fold_item_biases(A, B, item_biases, n_items, dtype)

    # ---- normalise storage: int32 CSR (the library validates and, if need be, sorts the rows), test values and factors in
    # `dtype`, row-major factors; nothing is copied that already has the right type and layout ----
    X_train = _csr_int32(X_train)
    if X_train.shape[0] > n_users:
        X_train = _csr_int32(X_train[:n_users])
    X_test = _csr_int32(X_test)
    test_values = X_test.data if X_test.data.dtype == dtype else X_test.data.astype(dtype)
    A, lda = _row_major_with_ld(A.astype(dtype, copy=False))
    B, ldb = _row_major_with_ld(B.astype(dtype, copy=False))
    c = np.ascontiguousarray

    want = {short: requested[name] for name, short, _ in _METRICS}
    outs = block = None
    if as_df and not cumulative:
        # the DataFrame's storage is allocated up front: one column-major [users, metrics] block whose columns ARE the output
        # arrays of the call -- pandas wraps it without a copy (building the frame from ten separate arrays consolidates them into
        # such a block: 0.9 ms for BASELINE C2's 138k users, 8 % of the call)
        keys = [key for (name, _, key) in _METRICS if requested[name]]
        block = np.empty((n_users, len(keys)), dtype=dtype, order="F")
        cols = iter(range(len(keys)))
        outs = [block[:, next(cols)] if requested[name] else np.empty(0, dtype=dtype) for name, _, _ in _METRICS]

    arrays = _binding.calc_metrics(
        A, lda, B, ldb, c(X_train.indptr), c(X_train.indices), c(X_test.indptr), c(X_test.indices), c(test_values),
        k, want, bool(cumulative), bool(break_ties_with_noise),
        bool(consider_cold_start), min_items_pool, min_pos_test, nthreads, seed, outs=outs)

    if block is not None:
        import pandas as pd
        if rename_k:
            keys = [key[:-1] + str(k) if key.endswith("@K") else key for key in keys]
        return pd.DataFrame(block, columns=keys, copy=False)
    out = {key: arr for (_, _, key), arr in zip(_METRICS, arrays) if arr.shape[0]}
    if not as_df:
        out["K"] = k
        return out
    return _as_frame(out, k, bool(cumulative), bool(rename_k))
