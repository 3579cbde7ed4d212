"""The oracle (oracle/recometrics_oracle.cpp) against the reference's golden vectors (CPU only).

Pins the CPU restatement to (a) the known-answer cases of the reference's own tests and (b) outputs
captured from the real reference (tests/golden/make_golden.py).  Bar: bit-exact.
"""
import numpy as np
import pytest

from _util import assert_same_bits, golden_cases, load_golden


@pytest.mark.parametrize("case", golden_cases())
def test_oracle_matches_reference_fixture(oracle, case):
    dtype, inp, variants = load_golden(case)
    assert variants
    for vi, (kw, expected) in enumerate(variants):
        got = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=dtype, **kw)
        assert set(got) == set(expected)
        for name in expected:
            assert_same_bits(got[name], expected[name], "%s v%d %s %s" % (case, vi, kw, name))


def test_known_answers_from_reference_tests(oracle):
    """Literal expectations of tests/testthat/test-ndcg.R:37-71,:107-124 and test-auc.R:22-61."""
    _, inp, variants = load_golden("g1_ndcg_neg_a")
    v = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **variants[0][0])["NDCG@K"][0]
    assert v > 0 and v == 0.7238540991261088
    _, inp, variants = load_golden("g1_ndcg_neg_b")
    v = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **variants[0][0])["NDCG@K"][0]
    assert v < 0 and v == -32.87200537510179
    _, inp, variants = load_golden("g1_ndcg_neg_c")
    v = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **variants[0][0])["NDCG@K"][0]
    assert v > 0 and v == 0.9211348612788796
    _, inp, variants = load_golden("g1_ndcg_fewer")
    v5 = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **variants[0][0])["NDCG@K"][0]
    v3 = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **variants[1][0])["NDCG@K"][0]
    want = (2 / np.log2(2) + 1 / np.log2(3)) / (3 / np.log2(2) + 2 / np.log2(3) + 1 / np.log2(4))
    assert v5 == v3 and abs(v5 - want) < 1e-15
    for case, roc in (("g2_auc_perfect", 1.0), ("g2_auc_zero", 0.0)):
        _, inp, variants = load_golden(case)
        for kw, _ in variants:
            r = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=np.float64, **kw)
            assert r["ROC_AUC"][0] == roc
            if roc == 1.0:
                assert r["PR_AUC"][0] == 1.0


def test_invalid_family_is_nan(oracle):
    """tests/testthat/test-ndcg.R:7-35,:73-105 => NA"""
    for case in golden_cases():
        if not case.startswith("g3_"):
            continue
        dtype, inp, variants = load_golden(case)
        for kw, _ in variants:
            r = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], dtype=dtype, **kw)
            assert np.isnan(r["NDCG@K"]).all(), case


def test_random_roc_auc_is_half(oracle):
    """tests/testthat/test-auc.R:7-20: random factors, no train data => mean ROC-AUC ~ 0.5 (tol 0.03)"""
    rng = np.random.default_rng(1)
    m, n, k = 100, 20, 3
    A = rng.standard_normal((m, k)).astype(np.float32)
    B = rng.standard_normal((n, k)).astype(np.float32)
    mask = rng.random((m, n)) < 0.1
    tep = np.concatenate([[0], np.cumsum(mask.sum(1))]).astype(np.int32)
    tei = np.nonzero(mask)[1].astype(np.int32)
    r = oracle.calc(A, B, (np.zeros(m + 1, np.int32), np.zeros(0, np.int32)), (tep, tei, None), k=5,
                    metrics=("roc",), noise=True)
    assert abs(np.nanmean(r["ROC_AUC"]) - 0.5) < 0.03


def test_thread_count_does_not_change_results(oracle):
    from recometrics_amd.synth import make_problem
    pr = make_problem(64, 500, 16, np.float32, mean_c=20, seed=5)
    a = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], k=6, nthreads=1)
    b = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], k=6, nthreads=4)
    for name in a:
        assert_same_bits(a[name], b[name], name)
