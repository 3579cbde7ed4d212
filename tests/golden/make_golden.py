#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- inputs + outputs of the REAL reference for the hot path.

Run in the build container only (needs /root/reference; it is compiled by oracle/Makefile into
oracle/_ref/librecometrics_ref.so -- canonical build).  The fixtures are
data: inputs and the reference's outputs.  No reference source is stored.

    python tests/golden/make_golden.py

Families:
  g1_ndcg_literal   literal cases of the reference's tests/testthat/test-ndcg.R:37-71,:107-124
  g2_auc            constructions of tests/testthat/test-auc.R:22-61 (ROC = 1 / 0, PR = 1)
  g3_invalid        test-ndcg.R:7-35,:73-105 "invalid" family => NaN
  g4_edge_users     n = 12 block: k_leq_n / only_ndcg / normal / no-test / cold users, single + cumulative
  g5_cum_ndcg       cumulative NDCG quirks (npos < K, negative and NaN test values)
  g6_random_*       random dense blocks, fp32 and fp64, all metrics, single + cumulative, noise off/on
  g7_cold_off       consider_cold_start = False
  g8_dyadic         23-bit dyadic factors (dot products exact in any order)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import METRICS, Reference  # noqa: E402
from recometrics_amd.synth import make_problem  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = Reference()


def csr_from_rows(rows, n, vals=None, dtype=np.float64):
    p = np.zeros(len(rows) + 1, dtype=np.int32)
    idx, v = [], []
    for u, r in enumerate(rows):
        order = np.argsort(r) if len(r) else []
        idx += [r[i] for i in order]
        if vals is not None:
            v += [vals[u][i] for i in order]
        p[u + 1] = p[u] + len(r)
    i = np.asarray(idx, dtype=np.int32)
    if vals is None:
        return p, i, np.ones(i.shape[0], dtype=dtype)
    return p, i, np.asarray(v, dtype=dtype)


def run_case(name, A, B, train, test, variants, dtype):
    """variants: list of dicts of calc kwargs; stores one output set per variant."""
    A = np.asarray(A, dtype=dtype)
    B = np.asarray(B, dtype=dtype)
    store = {"A": A, "B": B, "train_p": train[0], "train_i": train[1],
             "test_p": test[0], "test_i": test[1], "test_v": np.asarray(test[2], dtype=dtype)}
    meta = []
    for vi, kw in enumerate(variants):
        kw = dict(kw)
        kw.setdefault("metrics", METRICS)
        res = REF.calc(A, B, train, (test[0], test[1], store["test_v"]), dtype=dtype, **kw)
        for mname, arr in res.items():
            store["v%d__%s" % (vi, mname)] = arr
        kw["metrics"] = list(kw["metrics"])
        meta.append(kw)
    store["meta"] = np.frombuffer(json.dumps({"dtype": np.dtype(dtype).name, "variants": meta}).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print("%-28s %7.1f KB  %d variants" % (name, os.path.getsize(path) / 1024, len(variants)))


def main():
    f64, f32 = np.float64, np.float32
    empty_train = lambda m: (np.zeros(m + 1, np.int32), np.zeros(0, np.int32))  # noqa: E731
    ndcg_only = dict(metrics=("ndcg",), k=5, noise=False)

    # ---- g1: literal NDCG cases (test-ndcg.R:37-71, :107-124) --------------------------------------
    te = csr_from_rows([[2, 3, 5, 7, 9]], 10, [[1, 2, -3, 4, 5]])
    run_case("g1_ndcg_neg_a", np.ones((1, 1)), np.array([[0, 0, 3, 4, 0, 6, 0, 8, 9, 10.]]).T, empty_train(1), te, [ndcg_only], f64)
    te2 = csr_from_rows([[2, 3, 5, 7, 9]], 10, [[1, 2, -300, 4, 5]])
    run_case("g1_ndcg_neg_b", np.ones((1, 1)), np.array([[0, 0, 3, 4, 0, 600, 0, 8, 9, 10.]]).T, empty_train(1), te2, [ndcg_only], f64)
    run_case("g1_ndcg_neg_c", np.ones((1, 1)), np.array([[0, 0, 3, 4, 0, -6, 0, 8, 9, 10.]]).T, empty_train(1), te2, [ndcg_only], f64)
    te3 = csr_from_rows([[2, 3, 7]], 10, [[1, 2, 3]])
    Bf = np.array([[0.25, 0.125, 3, 4, 0.5, 0.0625, 0.75, 0.03125, 2, 1.]]).T   # tie-free variant of test-ndcg.R:115
    run_case("g1_ndcg_fewer", np.ones((1, 1)), Bf, empty_train(1), te3,
             [dict(metrics=("ndcg",), k=5, noise=False), dict(metrics=("ndcg",), k=3, noise=False)], f64)

    # ---- g2: perfect / zero AUC (test-auc.R:22-61) --------------------------------------------------
    rng = np.random.default_rng(1)
    n, npos = 100, 20
    pos = np.sort(rng.permutation(n)[:npos])
    Bp = np.full((n, 2), -100.0); Bp[pos] = 100.0; Bp += rng.standard_normal((n, 2))
    te = csr_from_rows([list(pos)], n)
    auc = dict(metrics=("roc", "pr"), k=10)
    run_case("g2_auc_perfect", np.ones((1, 2)), Bp, empty_train(1), te, [dict(auc, noise=False), dict(auc, noise=True)], f64)
    run_case("g2_auc_zero", np.ones((1, 2)), -Bp, empty_train(1), te, [dict(auc, noise=False), dict(auc, noise=True)], f64)
    run_case("g2_auc_perfect_f32", np.ones((1, 2)), Bp, empty_train(1), te, [dict(auc, noise=False)], f32)

    # ---- g3: invalid family (test-ndcg.R:7-35, :73-105) ---------------------------------------------
    te = csr_from_rows([[1, 4, 6, 8]], 10, [[0.5, -1.2, 2.0, 0.7]])
    A5 = np.ones((1, 5))
    both = [dict(ndcg_only, noise=True), dict(ndcg_only, noise=False)]
    run_case("g3_B_zero", A5, np.zeros((10, 5)), empty_train(1), te, both, f64)
    run_case("g3_B_ones", A5, np.ones((10, 5)), empty_train(1), te, both, f64)
    run_case("g3_B_nan", A5, np.full((10, 5), np.nan), empty_train(1), te, [dict(ndcg_only, noise=True)], f64)
    run_case("g3_B_inf", A5, np.full((10, 5), np.inf), empty_train(1), te, both, f64)
    Bn = rng.standard_normal((10, 5)); Bn[0, 0] = np.nan; Bn[2, 2] = np.nan
    run_case("g3_B_some_nan", A5, Bn, empty_train(1), te, [dict(ndcg_only, noise=True)], f64)
    Bi = Bn.copy(); Bi[0, 0] = -np.inf; Bi[2, 2] = np.inf
    run_case("g3_B_pm_inf", A5, Bi, empty_train(1), te, both, f64)
    Bok = rng.standard_normal((10, 5))
    te_neg = csr_from_rows([[1, 4, 6, 8]], 10, [[-0.5, -1.2, -2.0, -0.7]])
    run_case("g3_vals_all_neg", rng.standard_normal((1, 5)), Bok, empty_train(1), te_neg, both, f64)
    te_zero = csr_from_rows([[1, 4, 6, 8]], 10, [[0, 0, 0, 0.]])
    run_case("g3_vals_all_zero", rng.standard_normal((1, 5)), Bok, empty_train(1), te_zero, both, f64)

    # ---- g4: edge users, n = 12 ---------------------------------------------------------------------
    n = 12
    train_rows = [[0, 5], [0, 1, 2, 3, 4, 5, 6, 7], [3], [1, 2], [], [2, 9]]
    test_rows = [[3, 8], [8, 9, 10, 11], [0, 7, 11], [], [4, 6], [0, 1, 3, 4, 5, 6, 7, 8, 10, 11]]
    vals = [[2, 1], [1, 3, 2, 5], [4, 2, 1], [], [1, 1], [3, 1, 2, 5, 4, 1, 1, 2, 6, 1]]
    A = rng.standard_normal((6, 4)); B = rng.standard_normal((n, 4))
    tr = csr_from_rows(train_rows, n)[:2]
    te = csr_from_rows(test_rows, n, vals)
    variants = []
    for cum in (False, True):
        for K in (10, 4, 2):
            for noise in (False, True):
                variants.append(dict(k=K, cumulative=cum, noise=noise, seed=3))
    variants.append(dict(k=10, cumulative=True, noise=False, metrics=("p", "tp", "r", "hit", "roc")))      # quirk Q4, walk not run
    variants.append(dict(k=10, cumulative=False, noise=False, metrics=("p", "ap", "ndcg")))
    variants.append(dict(k=4, cumulative=True, noise=False, metrics=("ndcg",)))
    variants.append(dict(k=4, cumulative=False, noise=False, cold=False))
    variants.append(dict(k=4, cumulative=False, noise=False, min_items_pool=11))
    variants.append(dict(k=4, cumulative=False, noise=False, min_pos_test=3))                                # quirk Q1
    run_case("g4_edge_users_f64", A, B, tr, te, variants, f64)
    run_case("g4_edge_users_f32", A, B, tr, te, variants, f32)

    # ---- g5: cumulative NDCG quirks -----------------------------------------------------------------
    n = 30
    test_rows = [[2, 9, 17], [1, 3, 5, 7, 11, 13, 20], [4, 8, 15, 16, 23, 29], [0, 10, 20, 25]]
    vals = [[3, 1, 2], [1, 2, -3, 4, 5, -1, 2], [2, -1, -2, 3, 1, -4], [1, np.nan, 2, 3]]
    A = rng.standard_normal((4, 6)); B = rng.standard_normal((n, 6))
    # make sure some positives rank in the top-K
    for u, r in enumerate(test_rows):
        B[r[0]] += 0.8 * A[u] / np.linalg.norm(A[u]); B[r[-1]] += 0.6 * A[u] / np.linalg.norm(A[u])
    te = csr_from_rows(test_rows, n, vals)
    tr = csr_from_rows([[5], [0, 2], [], [7, 8, 9]], n)[:2]
    variants = [dict(k=K, cumulative=c, noise=False) for K in (5, 8) for c in (True, False)]
    run_case("g5_cum_ndcg_f64", A, B, tr, te, variants, f64)
    run_case("g5_cum_ndcg_f32", A, B, tr, te, variants, f32)

    # ---- g6: random dense blocks --------------------------------------------------------------------
    pr = make_problem(160, 1500, 32, f32, mean_c=40, seed=11)
    variants = [dict(k=10, cumulative=c, noise=nz, seed=7) for c in (False, True) for nz in (False, True)]
    variants.append(dict(k=1, cumulative=False, noise=False))
    variants.append(dict(k=37, cumulative=True, noise=False, metrics=("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr")))
    run_case("g6_random_f32", pr["A"], pr["B"], pr["train"], pr["test"], variants, f32)
    pr = make_problem(120, 900, 19, f64, mean_c=30, seed=12)
    run_case("g6_random_f64", pr["A"], pr["B"], pr["train"], pr["test"], variants, f64)

    # ---- g7: cold start off -------------------------------------------------------------------------
    pr = make_problem(40, 300, 8, f32, mean_c=6, seed=13)
    trp, tri = pr["train"]
    rows = [[] if u in (0, 7, 19) else list(tri[trp[u]:trp[u + 1]]) for u in range(40)]   # a few cold users
    newp, newi, _ = csr_from_rows(rows, 300)
    run_case("g7_cold_off_f32", pr["A"], pr["B"], (newp, newi), pr["test"],
             [dict(k=5, cold=False, noise=False), dict(k=5, cold=True, noise=False)], f32)

    # ---- g8: dyadic factors (order-independent dots) ------------------------------------------------
    m, n, k = 50, 700, 24
    A = rng.integers(-64, 65, (m, k)) / 64.0
    B = rng.integers(-2048, 2049, (n, k)) / 4096.0 + rng.integers(0, 2, (n, k)) / 65536.0
    pr = make_problem(m, n, k, f32, mean_c=25, seed=14)
    run_case("g8_dyadic_f32", A, B, pr["train"], pr["test"],
             [dict(k=10, noise=False), dict(k=10, noise=False, cumulative=True, metrics=("p", "ap", "ndcg", "rr"))], f32)


if __name__ == "__main__":
    main()
