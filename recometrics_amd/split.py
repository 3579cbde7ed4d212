"""Train / test splitting of implicit-feedback data -- host-side companion of the metric hot path ("next" row N3).

``split_reco_train_test`` follows ``recometrics.split_reco_train_test`` (reference recometrics/__init__.py:630-851):
same keywords, defaults, errors / warnings and return tuples; the work itself is ``rm_split_*`` of the C-ABI
(recometrics_amd/csrc/rm_split.cpp, CPU only)."""
from warnings import warn

import numpy as np

from . import _binding


_MODES = {"all": 0, "separated": 1, "joined": 2}


def _users_to_take(n_users, fraction, cap):
    """number of test users (reference recometrics/__init__.py:805-815, warnings included)"""
    if fraction is None:
        if cap > n_users:
            warn("'max_test_users' is larger than number of users. Will take all.")
        return min(cap, n_users)
    want = n_users * float(fraction)
    if want < 1:
        warn("Desired fraction of test users implies <1, will select 1 user.")
        want = 1
    return min(round(want), cap)


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

    n_users, n_items = X.shape
    mode = _MODES.get(split_type)
    if not max_test_users:                                # None / 0: no cap
        max_test_users = n_users
    # argument domain (the reference asserts these, recometrics/__init__.py:771-790): one table, first violation raises
    in_unit = lambda f: 0 < f < 1                          # noqa: E731
    for ok in (max_test_users > 0, seed >= 0, min_pos_test >= 0, min_items_pool >= 0,
               users_test_fraction is None or in_unit(users_test_fraction), in_unit(items_test_fraction), mode is not None):
        assert ok
    max_test_users, seed, min_pos_test, min_items_pool = (int(v) for v in (max_test_users, seed, min_pos_test, min_items_pool))
    # value errors, in the reference's order and wording (:795-826)
    for bad, message in (
        (min_pos_test >= n_items, "'min_pos_test' must be smaller than the number of columns in 'X'."),
        (min_items_pool >= n_items, "'min_items_pool' must be smaller than the number of columns in 'X'."),
        (mode != 0 and n_users < 2, "'X' has less than 2 rows."),
    ):
        if bad:
            raise ValueError(message)
    n_take = _users_to_take(n_users, users_test_fraction, max_test_users) if mode != 0 else 0

    X = _sorted_csr_int32(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    for bad, message in ((0 in X.shape, "'X' cannot be empty."), (X.data.shape[0] == 0, "'X' contains no non-zero entries.")):
        if bad:
            raise ValueError(message)

    res = _binding.split_csr(X.indptr, X.indices, X.data, n_items, mode, int(n_take), float(items_test_fraction),
                             bool(consider_cold_start), min_items_pool, min_pos_test, seed)
    if mode == 0:
        return _csr(res["train"], n_items), _csr(res["test"], n_items)
    if mode == 1:
        return _csr(res["rem"], n_items), _csr(res["train"], n_items), _csr(res["test"], n_items), res["users_test"]
    return _csr(res["train"], n_items), _csr(res["test"], n_items), res["users_test"]
