"""CPU-only: the library's CSR normalisation (csrc/rm_csr.cpp -- rm_csr_rows_sorted / rm_csr_sort_rows, what the Python API runs in
front of the metric call instead of SciPy's single-threaded `sort_indices()`, reference recometrics/__init__.py:35-41) against
SciPy itself: same answer to "are the rows sorted", same arrays after the sort (stable for repeated columns), any thread count,
ragged and empty rows; and `_sorted_csr_int32` leaves a matrix exactly as the reference's `_as_csr` + `_cast_indices_to_int32` would."""
import numpy as np
import pytest
import scipy.sparse as sp

from recometrics_amd import _binding as hip


def _random_csr(m, n, density, seed, dtype=np.float32, shuffle=0.5, dup=False):
    rng = np.random.default_rng(seed)
    X = sp.random(m, n, density=density, format="csr", random_state=seed, dtype=np.float64).astype(dtype)
    X.sort_indices()
    ip, ix, d = X.indptr.astype(np.int32), X.indices.astype(np.int32).copy(), X.data.copy()
    for r in range(m):
        a, e = ip[r], ip[r + 1]
        if e - a > 1 and rng.random() < shuffle:
            perm = rng.permutation(e - a)
            ix[a:e], d[a:e] = ix[a:e][perm], d[a:e][perm]
        if dup and e - a > 2:
            ix[a + 1] = ix[a]                                      # a repeated column (not canonical, but legal CSR)
    return ip, ix, d


@pytest.mark.parametrize("nthreads", [1, 3, 0])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_check_and_sort_equal_scipy(nthreads, dtype):
    m, n = 5000, 700
    ip, ix, d = _random_csr(m, n, 0.01, 1, dtype)
    ref = sp.csr_array((d.copy(), ix.copy(), ip.copy()), shape=(m, n))
    assert ref.has_sorted_indices == hip.csr_rows_sorted(ip, ix, nthreads) == False      # noqa: E712
    ref.sort_indices()
    hip.csr_sort_rows(ip, ix, d, nthreads)
    assert (ix == ref.indices).all() and (d == ref.data).all()
    assert hip.csr_rows_sorted(ip, ix, nthreads)


def test_sort_is_stable_for_repeated_columns_and_handles_edges():
    ip, ix, d = _random_csr(300, 50, 0.2, 2, np.float64, shuffle=1.0, dup=True)
    want_ix, want_d = ix.copy(), d.copy()
    for r in range(300):
        a, e = ip[r], ip[r + 1]
        o = np.argsort(want_ix[a:e], kind="stable")
        want_ix[a:e], want_d[a:e] = want_ix[a:e][o], want_d[a:e][o]
    hip.csr_sort_rows(ip, ix, d, 2)
    assert (ix == want_ix).all() and (d == want_d).all()
    # equal neighbours count as sorted (SciPy's has_sorted_indices), empty matrices and empty rows are fine, no values is fine
    assert hip.csr_rows_sorted(np.array([0, 2, 2, 4], np.int32), np.array([3, 3, 0, 9], np.int32))
    assert not hip.csr_rows_sorted(np.array([0, 2, 2, 4], np.int32), np.array([3, 1, 0, 9], np.int32))
    assert hip.csr_rows_sorted(np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert hip.csr_rows_sorted(np.zeros(5, np.int32), np.zeros(0, np.int32))
    only_idx = np.array([5, 1, 3, 2, 0], np.int32)
    hip.csr_sort_rows(np.array([0, 3, 5], np.int32), only_idx, None, 1)
    assert only_idx.tolist() == [1, 3, 5, 0, 2]


def test_large_input_in_parallel_equals_one_thread():
    ip, ix, d = _random_csr(60000, 3000, 0.006, 3, np.float32, shuffle=0.3)          # > 2^18 entries per thread: the parallel path
    assert ip[-1] > 3 * (1 << 18)
    a_ix, a_d = ix.copy(), d.copy()
    hip.csr_sort_rows(ip, a_ix, a_d, 1)
    hip.csr_sort_rows(ip, ix, d, 0)
    assert (ix == a_ix).all() and (d == a_d).all() and hip.csr_rows_sorted(ip, ix, 0) and hip.csr_rows_sorted(ip, ix, 1)


def test_api_normalisation_matches_the_reference_recipe():
    """recometrics_amd._sorted_csr_int32 == the reference's `_as_csr` (tocsr + sort_indices, in place) + `_cast_indices_to_int32`"""
    from recometrics_amd import _sorted_csr_int32
    m, n = 400, 90
    ip, ix, d = _random_csr(m, n, 0.08, 4, np.float64)
    for idx_dtype in (np.int32, np.int64):
        X = sp.csr_array((d.copy(), ix.astype(idx_dtype), ip.astype(idx_dtype)), shape=(m, n))
        want = sp.csr_array((d.copy(), ix.astype(idx_dtype), ip.astype(idx_dtype)), shape=(m, n))
        want.sort_indices()
        got = _sorted_csr_int32(X, 2)
        assert got.indices.dtype == np.int32 and got.indptr.dtype == np.int32 and got.has_sorted_indices
        assert (got.indices == want.indices).all() and (got.data == want.data).all() and (got.indptr == want.indptr).all()
        if idx_dtype == np.int32:
            assert got is X and (X.indices == want.indices).all()                 # sorted in place, like the reference
        assert _sorted_csr_int32(got, 2) is got                                    # the flag is set: nothing to do
    coo = sp.coo_array((d, (np.repeat(np.arange(m), np.diff(ip)), ix)), shape=(m, n))
    got = _sorted_csr_int32(coo, 1)
    assert got.format == "csr" and got.has_sorted_indices and (got.toarray() == want.toarray()).all()


def test_metric_call_normalisation_leaves_the_rows_alone():
    """recometrics_amd._csr_int32 (what calc_reco_metrics hands to the library): int32 index arrays, rows in the caller's order,
    the caller's matrix never modified -- the library validates the rows on the device and sorts a copy when it has to
    (tests/test_hip_csr_validation.py)"""
    from recometrics_amd import _csr_int32
    m, n = 300, 70
    ip, ix, d = _random_csr(m, n, 0.1, 9, np.float32)
    for idx_dtype in (np.int32, np.int64):
        X = sp.csr_array((d.copy(), ix.astype(idx_dtype), ip.astype(idx_dtype)), shape=(m, n))
        before = (X.indices.copy(), X.data.copy(), X.indptr.copy())
        got = _csr_int32(X)
        assert got.indices.dtype == np.int32 and got.indptr.dtype == np.int32
        assert (got.indices == ix).all() and (got.data == d).all() and (got.indptr == ip).all()      # unsorted as given
        assert (X.indices == before[0]).all() and (X.data == before[1]).all() and (X.indptr == before[2]).all()
        if idx_dtype == np.int32:
            assert got is X
    coo = sp.coo_array((d, (np.repeat(np.arange(m), np.diff(ip)), ix)), shape=(m, n))
    assert _csr_int32(coo).format == "csr"


def test_scipy_still_has_the_cached_sortedness_attribute():
    """_sorted_csr_int32 (the split's normalisation) trusts SciPy's cached answer through the private attribute behind the public
    `has_sorted_indices` property; if a SciPy release renames it the code stays correct but pays for a full check per call --
    this test says so instead of letting it happen silently"""
    X = sp.csr_array(np.array([[0, 2, 1], [3, 0, 0]], dtype=np.float64))
    X.has_sorted_indices = True
    assert getattr(X, "_has_sorted_indices", None) is True
