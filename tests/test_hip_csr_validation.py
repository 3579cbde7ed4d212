"""GPU: the CSR inputs are validated on the device, in front of everything that indexes by them.

The reference's callers guarantee sorted rows (recometrics/__init__.py:35-41 sorts X_train / X_test with SciPy at :553-558)
and nobody range-checks the indices: on the CPU a bad index is a segfault.  Here the rows are uploaded as they come;
the plan kernels (csrc/rm_prep.hpp k_check_csr_ptr / k_check_csr_rows) verify "index pointers monotone and inside the arrays,
every index in [0, n), every row ascending".  Unsorted rows are sorted by the library (a copy, host side) and the call runs
again -- same outputs as for the sorted matrix; anything else is RM_ERR_INVALID (ValueError) with a message, never a fault."""
import numpy as np
import pytest

from _util import assert_same_bits
from test_hip_parity import hip, hip_calc  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _problem(m=900, n=5000, k=32, dtype=np.float32, mean_c=70, seed=11):
    from recometrics_amd.synth import make_problem
    return make_problem(m, n, k, dtype, mean_c=mean_c, seed=seed)


def _shuffled_rows(p, idx, val, rng, frac=0.5):
    """the same matrix with the entries of about `frac` of the rows in random order (values travel with their indices)"""
    idx, val = idx.copy(), None if val is None else val.copy()
    for u in range(p.shape[0] - 1):
        a, b = p[u], p[u + 1]
        if b - a > 1 and rng.random() < frac:
            perm = rng.permutation(b - a)
            idx[a:b] = idx[a:b][perm]
            if val is not None:
                val[a:b] = val[a:b][perm]
    return idx, val


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("which", ["train", "test", "both"])
@pytest.mark.parametrize("noise", [False, True])
def test_unsorted_rows_give_the_outputs_of_the_sorted_matrix(hip, dtype, which, noise):
    pr = _problem(dtype=dtype)
    rng = np.random.default_rng(3)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    want = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype, noise=noise, seed=5)
    tri2, tei2, tev2 = tri, tei, tev
    if which in ("train", "both"):
        tri2, _ = _shuffled_rows(trp, tri, None, rng)
    if which in ("test", "both"):
        tei2, tev2 = _shuffled_rows(tep, tei, tev, rng)
    keep = (tri2.copy(), tei2.copy(), tev2.copy())
    got = hip_calc(hip, pr["A"], pr["B"], (trp, tri2), (tep, tei2, tev2), 10, dtype=dtype, noise=noise, seed=5)
    for name in want:
        assert_same_bits(got[name], want[name], "%s rows unsorted: %s" % (which, name))
    # the caller's arrays are const: the library sorted a copy
    assert (tri2 == keep[0]).all() and (tei2 == keep[1]).all() and (tev2 == keep[2]).all()


def test_unsorted_rows_in_a_later_user_batch(hip, monkeypatch):
    """host entry in user batches: the defect sits in the LAST batch only (earlier batches have already handed their outputs
    over when it is found); the call still returns the sorted matrix's outputs for every user"""
    pr = _problem(m=5000, n=3000, mean_c=40)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    want = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 7)
    monkeypatch.setenv("RM_BATCH_USERS", "1024")
    u = next(v for v in range(4800, 5000) if tep[v + 1] - tep[v] >= 2)
    tei2, tev2 = tei.copy(), tev.copy()
    a, b = tep[u], tep[u + 1]
    tei2[a:b] = tei2[a:b][::-1]; tev2[a:b] = tev2[a:b][::-1]
    got = hip_calc(hip, pr["A"], pr["B"], (trp, tri), (tep, tei2, tev2), 7)
    for name in want:
        assert_same_bits(got[name], want[name], name)


def test_unsorted_rows_through_the_python_api(hip):
    """calc_reco_metrics with SciPy matrices whose rows are not sorted (has_sorted_indices unknown or False): the reference's
    outputs for the sorted matrices, and -- unlike the reference -- the caller's matrices untouched"""
    import scipy.sparse as sp
    from recometrics_amd import calc_reco_metrics
    pr = _problem(m=600, n=4000)
    rng = np.random.default_rng(9)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    m, n = pr["A"].shape[0], pr["B"].shape[0]
    Xtr = sp.csr_array((np.ones(tri.shape[0], np.float32), tri, trp), shape=(m, n))
    Xte = sp.csr_array((tev, tei, tep), shape=(m, n))
    want = calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=10, all_metrics=True, as_df=False)
    tri2, _ = _shuffled_rows(trp, tri, None, rng)
    tei2, tev2 = _shuffled_rows(tep, tei, tev, rng)
    Xtr2 = sp.csr_array((np.ones(tri.shape[0], np.float32), tri2, trp), shape=(m, n))
    Xte2 = sp.csr_array((tev2, tei2, tep), shape=(m, n))
    before = (Xtr2.indices.copy(), Xte2.indices.copy(), Xte2.data.copy())
    got = calc_reco_metrics(Xtr2, Xte2, pr["A"], pr["B"], k=10, all_metrics=True, as_df=False)
    for name in want:
        if name != "K":
            assert_same_bits(got[name], want[name], name)
    assert (Xtr2.indices == before[0]).all() and (Xte2.indices == before[1]).all() and (Xte2.data == before[2]).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("where", ["train", "test"])
@pytest.mark.parametrize("value", ["n", "n+1000000", "-1", "int_min"])
def test_out_of_range_index_is_an_error_not_a_fault(hip, dtype, where, value):
    pr = _problem(dtype=dtype)
    n = pr["B"].shape[0]
    bad = {"n": n, "n+1000000": n + 1000000, "-1": -1, "int_min": -2 ** 31}[value]
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    tri, tei = tri.copy(), tei.copy()
    (tri if where == "train" else tei)[(trp if where == "train" else tep)[500]] = bad          # first entry of row 500
    with pytest.raises(ValueError, match="out of range"):
        hip_calc(hip, pr["A"], pr["B"], (trp, tri), (tep, tei, tev), 10, dtype=dtype)
    assert "row 500" in hip.load().rm_last_error().decode()
    # ... and the library is fine afterwards
    want = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype)
    assert np.isfinite(want["P@K"]).any()


@pytest.mark.parametrize("where", ["train", "test"])
@pytest.mark.parametrize("defect", ["decreasing", "negative", "beyond"])
def test_bad_index_pointers_are_an_error(hip, where, defect):
    pr = _problem()
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    p = (trp if where == "train" else tep).copy()
    if defect == "decreasing":
        p[301] = p[300] - 1 if p[300] > 0 else p[302] + 5
    elif defect == "negative":
        p[0] = -3
    else:
        p[-1] = p[-1]            # host entry: the last pointer IS the length; the device entry covers "beyond" below
        p[400] = p[-1] + 7
    with pytest.raises(ValueError, match="index pointers"):
        hip_calc(hip, pr["A"], pr["B"], (p, tri) if where == "train" else (trp, tri), (tep, tei, tev) if where == "train" else (p, tei, tev), 10)


_DEVICE_SCRIPT = r"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
from _util import same_bits
torch.cuda.set_device(0); hip.load(); hip.set_device(0)
dev = torch.device("cuda", 0)
pr = make_problem(700, 4000, 32, np.float32, mean_c=60, seed=21)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
A, B = t(pr["A"]), t(pr["B"])
def run(trp, tri, tep, tei, tev, nnz_tr=None, nnz_te=None):
    d = [t(x) for x in (trp, tri, tep, tei, tev)]
    out = torch.full((10, 700), -7.0, dtype=torch.float32, device=dev)
    hip.calc_metrics_device(np.float32, A.data_ptr(), 32, B.data_ptr(), 32, 700, 4000, 32, d[0].data_ptr(), d[1].data_ptr(),
                            tri.shape[0] if nnz_tr is None else nnz_tr, d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                            tei.shape[0] if nnz_te is None else nnz_te, 10, [out[i].data_ptr() for i in range(10)])
    torch.cuda.synchronize()
    return out.cpu().numpy()
res = {}
want = run(trp, tri, tep, tei, tev)
rng = np.random.default_rng(1)
tei2, tev2, tri2 = tei.copy(), tev.copy(), tri.copy()
for u in range(0, 700, 3):
    a, b = tep[u], tep[u + 1]
    perm = rng.permutation(b - a); tei2[a:b] = tei2[a:b][perm]; tev2[a:b] = tev2[a:b][perm]
    a, b = trp[u], trp[u + 1]
    tri2[a:b] = tri2[a:b][rng.permutation(b - a)]
got = run(trp, tri2, tep, tei2, tev2)
res["unsorted_same_bits"] = bool(same_bits(got, want).all())
def fails(*a, **k):
    try:
        run(*a, **k)
    except ValueError as e:
        return str(e)
    return None
bad = tei.copy(); bad[tep[123]] = 4000
res["index_n"] = fails(trp, tri, tep, bad, tev)
bad = tri.copy(); bad[trp[77]] = -5
res["index_negative"] = fails(trp, bad, tep, tei, tev)
p = tep.copy(); p[200] = p[199] - 1
res["ptr_decreasing"] = fails(trp, tri, p, tei, tev)
res["ptr_beyond"] = fails(trp, tri, tep, tei, tev, nnz_te=int(tep[-1]) - 1)
res["ok_again"] = bool(same_bits(run(trp, tri, tep, tei, tev), want).all())
print(json.dumps(res))
"""


def test_device_pointer_entry_validates_too(hip):
    """rm_calc_metrics_dev_*: the same checks on arrays that live in HBM already (torch is only the memory owner: a process of
    its own, torch loaded before the library); unsorted rows are sorted through a copy in the library's workspace"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _DEVICE_SCRIPT % {"root": root}], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert got["unsorted_same_bits"] and got["ok_again"], got
    assert got["index_n"] and "out of range" in got["index_n"] and "row 123" in got["index_n"], got
    assert got["index_negative"] and "out of range" in got["index_negative"], got
    assert got["ptr_decreasing"] and "index pointers" in got["ptr_decreasing"], got
    assert got["ptr_beyond"] and "index pointers" in got["ptr_beyond"], got


def test_rank_entry_refuses_unsorted_rows(hip):
    """rm_rank_*: pos_rank is indexed by the caller's entry order, so there is nothing to sort behind the caller's back"""
    pr = _problem(m=200, n=2000)
    trp, tri = pr["train"]; tep, tei, _ = pr["test"]
    tei2 = tei.copy()
    a, b = tep[10], tep[11]
    tei2[a:b] = tei2[a:b][::-1]
    with pytest.raises(ValueError, match="sorted"):
        hip.rank(pr["A"], pr["B"], trp, tri, tep, tei2, 5)
