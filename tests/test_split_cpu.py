"""Host-only code, run on the CPU here and once more on the GPU box (the `gpu`-marked twin at the end: the same checks through
the library that ships there): rm_split_* (recometrics_amd/csrc/rm_split.cpp) against fixtures captured from the real reference's
split functions (tests/golden/make_golden_split.py), bit for bit; plus the invariants of the reference's own
tests/testthat/test-split.R:7-93 (X_train + X_test == X[users_test], row counts, error on impossible criteria)."""
import glob
import json
import os

import numpy as np
import pytest
from scipy.sparse import csr_array

from recometrics_amd import _binding, split_reco_train_test

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["train_p", "train_i", "train_v", "test_p", "test_i", "test_v", "rem_p", "rem_i", "rem_v", "users_test"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "split", "*.npz"))))
def test_split_matches_reference_fixture(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    n = int(z["shape"][1])
    for dtype in (np.float64, np.float32):
        for vi, kw in enumerate(meta):
            args = dict(n_users_test=kw.get("n_users_test", 0), test_fraction=kw["frac"], consider_cold_start=kw.get("cold", False),
                        min_items_pool=kw.get("min_items_pool", 2), min_pos_test=kw.get("min_pos_test", 1), seed=kw["seed"])
            if kw["raised"]:
                with pytest.raises(RuntimeError):
                    _binding.split_csr(z["X_p"], z["X_i"], z["X_v"].astype(dtype), n, kw["mode"], **args)
                continue
            got = _binding.split_csr(z["X_p"], z["X_i"], z["X_v"].astype(dtype), n, kw["mode"], **args)
            flat = {"users_test": got["users_test"]}
            for part in ("train", "test", "rem"):
                flat[part + "_p"], flat[part + "_i"], flat[part + "_v"] = got[part]
            for nm in NAMES:
                want = z["v%d__%s" % (vi, nm)]
                if kw["mode"] != 1 and nm.startswith("rem_"):
                    continue
                assert flat[nm].shape == want.shape, (path, vi, nm)
                assert (flat[nm] == want.astype(flat[nm].dtype)).all(), (path, vi, nm)


def test_split_invariants_like_the_reference_tests():
    rng = np.random.default_rng(3)
    X = csr_array((rng.random((120, 70)) < 0.15) * rng.integers(1, 6, (120, 70)).astype(np.float64))
    Xtr, Xte = split_reco_train_test(X, split_type="all", items_test_fraction=0.4, seed=5)
    assert Xtr.shape == X.shape and Xte.shape == X.shape and ((Xtr + Xte) != X).nnz == 0
    Xrem, Xtr, Xte, users = split_reco_train_test(X, split_type="separated", users_test_fraction=0.25, seed=5)
    assert Xtr.shape[0] == Xte.shape[0] == users.shape[0] and Xrem.shape[0] == X.shape[0] - users.shape[0]
    assert ((Xtr + Xte) != X[users]).nnz == 0
    rest = np.setdiff1d(np.arange(X.shape[0]), users)
    assert (Xrem != X[rest]).nnz == 0
    Xtr2, Xte2, users2 = split_reco_train_test(X, split_type="joined", users_test_fraction=0.25, seed=5)
    assert (users2 == users).all() and Xtr2.shape[0] == X.shape[0] and (Xte2 != Xte).nnz == 0
    assert (Xtr2[:users.shape[0]] != Xtr).nnz == 0 and (Xtr2[users.shape[0]:] != Xrem).nnz == 0
    with pytest.raises(RuntimeError):                         # nobody can have 60 test positives out of 70 items
        split_reco_train_test(X, split_type="separated", min_pos_test=60, seed=5)
    with pytest.raises(ValueError):
        split_reco_train_test(X, split_type="separated", min_items_pool=70)


@pytest.mark.gpu
def test_split_through_the_shipped_library_on_the_gpu_box():
    """N3 is host code, but it lives in the library that travels to the GPU box: run the fixture and invariant checks there once."""
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "split", "*.npz"))):
        test_split_matches_reference_fixture(path)
    test_split_invariants_like_the_reference_tests()
