"""G9: recometrics_amd.calc_reco_metrics against what the REFERENCE's Python API returned for the same inputs
(tests/golden/api/, captured by tests/golden/make_golden_api.py from the imported reference package): dict keys and their
order, K, shapes, dtypes, values, DataFrame columns / dtypes, warnings, exceptions.  The factors are dyadic, so the values are
independent of the reference build's summation order and are compared bit for bit (ROC-AUC: x87 long double there, 1e-5)."""
import json
import os
import warnings

import numpy as np
import pytest
from scipy.sparse import csr_matrix

from _util import assert_close, assert_same_bits

API_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api")
INDEX = json.load(open(os.path.join(API_DIR, "api.json")))


def _inputs(entry):
    z = np.load(os.path.join(API_DIR, entry["name"] + ".npz"))
    m, n = entry["shape"]
    Xtr = csr_matrix((z["trv"], z["tri"], z["trp"]), shape=(m, n))
    Xte = csr_matrix((z["tev"], z["tei"], z["tep"]), shape=(m, n))
    B = np.asfortranarray(z["B"]) if entry["fortran_b"] else z["B"]
    kw = dict(entry["kwargs"])
    if entry["has_item_biases"]:
        kw["item_biases"] = z["item_biases"]
    return z, Xtr, Xte, z["A"], B, kw


@pytest.mark.parametrize("entry", [e for e in INDEX if e["error"]], ids=lambda e: e["name"])
def test_reference_exceptions(entry):
    """argument errors are raised before anything touches the device: same exception type and message"""
    from recometrics_amd import calc_reco_metrics
    _, Xtr, Xte, A, B, kw = _inputs(entry)
    exc = {"ValueError": ValueError, "AssertionError": AssertionError, "TypeError": TypeError}[entry["error"][0]]
    with pytest.raises(exc) as ei:
        calc_reco_metrics(Xtr, Xte, A, B, **kw)
    assert str(ei.value) == entry["error"][1]


@pytest.mark.gpu
@pytest.mark.parametrize("entry", [e for e in INDEX if not e["error"]], ids=lambda e: e["name"])
def test_reference_api_outputs(entry):
    from recometrics_amd import calc_reco_metrics
    z, Xtr, Xte, A, B, kw = _inputs(entry)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        d = calc_reco_metrics(Xtr.copy(), Xte.copy(), A, B, as_df=False, **kw)
        df = calc_reco_metrics(Xtr.copy(), Xte.copy(), A, B, as_df=True, **kw)
    assert sorted({str(w.message) for w in wlist}) == entry["warnings"]
    assert list(d.keys()) == entry["dict_keys"]
    assert d["K"] == entry["K"]
    for key in entry["dict_keys"]:
        if key == "K":
            continue
        want = z["out__" + key]
        got = np.asarray(d[key])
        assert got.dtype == want.dtype and got.shape == want.shape, (key, got.dtype, want.dtype, got.shape, want.shape)
        assert_close(got, want, 1e-5, key)
        if key != "ROC_AUC":
            assert_same_bits(got, want, key + " (bitwise)")
    if not (kw.get("cumulative") and any(c in entry["dict_keys"] for c in ("ROC_AUC", "PR_AUC"))):
        assert [str(c) for c in df.columns] == entry["df_columns"]
        assert [str(t) for t in df.dtypes] == entry["df_dtypes"]
        assert list(df.shape) == entry["df_shape"]


@pytest.mark.gpu
@pytest.mark.parametrize("entry", [e for e in INDEX if not e["error"]][:4], ids=lambda e: e["name"])
def test_unsorted_and_wide_index_matrices_give_the_reference_outputs(entry):
    """the reference sorts the column indices of X_train / X_test itself (recometrics/__init__.py:35-41); here that pass runs in the
    library (csrc/rm_csr.cpp).  Rows shuffled, int64 index arrays, SciPy's sortedness flag unknown: same outputs, bit for bit"""
    from scipy.sparse import csr_array
    from recometrics_amd import calc_reco_metrics
    z, Xtr, Xte, A, B, kw = _inputs(entry)
    rng = np.random.default_rng(5)

    def scrambled(X):
        ip, ix, d = X.indptr.astype(np.int64), X.indices.astype(np.int64).copy(), X.data.copy()
        for r in range(X.shape[0]):
            a, e = ip[r], ip[r + 1]
            perm = rng.permutation(e - a)
            ix[a:e], d[a:e] = ix[a:e][perm], d[a:e][perm]
        return csr_array((d, ix, ip), shape=X.shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = calc_reco_metrics(scrambled(Xtr), scrambled(Xte), A, B, as_df=False, **kw)
    for key in entry["dict_keys"]:
        if key == "K":
            continue
        want, got = z["out__" + key], np.asarray(d[key])
        assert_close(got, want, 1e-5, key)
        if key != "ROC_AUC":
            assert_same_bits(got, want, key + " (bitwise, scrambled input)")
