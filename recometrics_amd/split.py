"""Train / test splitting of implicit-feedback data -- host-side companion of the metric hot path ("next" row N3).

``split_reco_train_test`` follows ``recometrics.split_reco_train_test`` (reference recometrics/__init__.py:630-851):
same keywords, defaults, errors / warnings and return tuples; the work itself is ``rm_split_*`` of the C-ABI
(recometrics_amd/csrc/rm_split.cpp, CPU only)."""
from warnings import warn

import numpy as np

from . import _binding


def _csr(parts, n_items):
    from scipy.sparse import csr_array
    indptr, indices, data = parts
    return csr_array((data, indices, indptr), shape=(max(indptr.shape[0] - 1, 0), n_items))


def split_reco_train_test(
    X,
    split_type="separated",
    users_test_fraction=0.1,
    max_test_users=10000,
    items_test_fraction=0.3,
    min_items_pool=2,
    min_pos_test=1,
    consider_cold_start=False,
    seed=1,
):
    """Create train-test splits of implicit-feedback data (CSR user-item interactions).

    ``split_type="all"``: every user's row is split -> ``(X_train, X_test)``.
    ``split_type="separated"``: test users are drawn at random among the eligible ones ->
    ``(X_rem, X_train, X_test, users_test)`` (train / test rows of the test users, the untouched other users).
    ``split_type="joined"``: -> ``(X_train, X_test, users_test)`` with ``X_train`` = train rows of the test users
    stacked on top of the other users.  See the reference's docstring (recometrics/__init__.py:641-766) for the
    meaning of every argument; results are identical for the same ``seed``."""
    from . import _sorted_csr_int32                       # CSR normalisation shared with calc_reco_metrics

    if not max_test_users:
        max_test_users = X.shape[0]
    assert max_test_users > 0 and seed >= 0 and min_pos_test >= 0 and min_items_pool >= 0
    max_test_users, seed = int(max_test_users), int(seed)
    min_pos_test, min_items_pool = int(min_pos_test), int(min_items_pool)
    if users_test_fraction is not None:
        assert 0 < users_test_fraction < 1
        users_test_fraction = float(users_test_fraction)
    assert 0 < items_test_fraction < 1
    items_test_fraction = float(items_test_fraction)
    assert split_type in ("all", "separated", "joined")

    n_users, n_items = X.shape
    if min_pos_test >= n_items:
        raise ValueError("'min_pos_test' must be smaller than the number of columns in 'X'.")
    if min_items_pool >= n_items:
        raise ValueError("'min_items_pool' must be smaller than the number of columns in 'X'.")

    n_take = 0
    if split_type != "all":
        if n_users < 2:
            raise ValueError("'X' has less than 2 rows.")
        if users_test_fraction is not None:
            n_take = n_users * users_test_fraction
            if n_take < 1:
                warn("Desired fraction of test users implies <1, will select 1 user.")
                n_take = 1
            n_take = min(round(n_take), max_test_users)
        else:
            if max_test_users > n_users:
                warn("'max_test_users' is larger than number of users. Will take all.")
            n_take = min(max_test_users, n_users)

    X = _sorted_csr_int32(X)
    if not X.shape[0] or not X.shape[1]:
        raise ValueError("'X' cannot be empty.")
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    if not X.data.shape[0]:
        raise ValueError("'X' contains no non-zero entries.")

    mode = {"all": 0, "separated": 1, "joined": 2}[split_type]
    res = _binding.split_csr(X.indptr, X.indices, X.data, n_items, mode, int(n_take), items_test_fraction,
                             bool(consider_cold_start), min_items_pool, min_pos_test, seed)
    if mode == 0:
        return _csr(res["train"], n_items), _csr(res["test"], n_items)
    if mode == 1:
        return _csr(res["rem"], n_items), _csr(res["train"], n_items), _csr(res["test"], n_items), res["users_test"]
    return _csr(res["train"], n_items), _csr(res["test"], n_items), res["users_test"]
