#!/usr/bin/env python3
"""Generates tests/golden/g9_noise_*.npz -- inputs + outputs of the REAL reference (oracle/_ref, canonical build) with
break_ties_with_noise=True on inputs where the noise DECIDES the ranking (reference src/recometrics.hpp:528-534).

Run in the build container only (needs /root/reference compiled by oracle/Makefile).  Fixtures are data.

  g9_noise_ties_f64   scores in clusters 1e-13 apart (far below the +-1e-12 noise): the order inside a cluster, hence the
                      top-K lists and the ranks of the test items, is whatever each user's mt19937(seed + user) stream says
  g9_noise_zone_f32   fp32: the noise only matters below |score| ~ 3e-5 -- blocks of exactly-zero scores (cold items with
                      zero factors), tiny scores that the noise shifts, test items inside them, top-K lists that reach into them
  g9_noise_cold_f32   random factors with 15 % of the items zeroed (the everyday form of the above), 64-bit seed
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import csr_from_rows, run_case  # noqa: E402
from recometrics_amd.synth import make_problem  # noqa: E402


def main():
    f64, f32 = np.float64, np.float32
    rng = np.random.default_rng(99)

    # ---- f64 near-ties ----
    m, n = 48, 420
    base = np.sort(rng.standard_normal(60))
    b = (base[:, None] + np.arange(7)[None, :] * 1e-13).reshape(-1)          # 60 clusters of 7 scores 1e-13 apart
    perm = rng.permutation(n)
    B = np.zeros((n, 2)); B[perm, 0] = b
    A = np.zeros((m, 2)); A[:, 0] = 1.0                                       # every user sees the same scores, its own noise
    A[m // 2:, 0] = -1.0                                                     # ... half of them in reverse
    test_rows, train_rows = [], []
    for u in range(m):
        items = rng.permutation(n)
        nte = int(rng.integers(3, 40)); ntr = int(rng.integers(0, 30))
        test_rows.append(sorted(items[:nte].tolist())); train_rows.append(sorted(items[nte:nte + ntr].tolist()))
    te = csr_from_rows(test_rows, n, [list(rng.integers(1, 9, len(r)).astype(float)) for r in test_rows])
    tr = csr_from_rows(train_rows, n)[:2]
    variants = [dict(k=10, cumulative=c, noise=True, seed=sd) for c in (False, True) for sd in (1, 12345678901)]
    variants.append(dict(k=10, cumulative=False, noise=False))
    run_case("g9_noise_ties_f64", A, B, tr, te, variants, f64)

    # ---- f32 zone ----
    m, n = 40, 600
    # (the tiny scores are 64+ ulps apart: the noise moves them by up to 9 ulps but cannot make two of them EQUAL -- the
    # reference's order of exactly tied scores is std::sort's internal business and is pinned by nothing; the zeros get
    # 2^24 distinct noise values and order by them)
    vals = np.concatenate([np.zeros(250), 1e-6 + np.arange(100) * 7.3e-12, -2e-6 - np.arange(100) * 1.5e-11,
                           rng.standard_normal(150) * 0.5]).astype(np.float32)
    perm = rng.permutation(n)
    B = np.zeros((n, 3), np.float32); B[perm, 1] = vals
    A = np.zeros((m, 3), np.float32); A[:, 1] = 1.0
    A[m // 2:, 1] = -1.0
    test_rows, train_rows = [], []
    for u in range(m):
        items = rng.permutation(n)
        nte = int(rng.integers(2, 90)); ntr = int(rng.integers(0, 60))
        test_rows.append(sorted(items[:nte].tolist())); train_rows.append(sorted(items[nte:nte + ntr].tolist()))
    te = csr_from_rows(test_rows, n, [list(rng.integers(1, 9, len(r)).astype(float)) for r in test_rows], dtype=f32)
    tr = csr_from_rows(train_rows, n)[:2]
    variants = [dict(k=K, cumulative=c, noise=True, seed=5) for K in (10, 100) for c in (False, True)]
    # (no noise-off variant: 250 exactly tied scores are then ordered by std::sort's internals -- the case the flag exists for)
    run_case("g9_noise_zone_f32", A, B, tr, te, variants, f32)

    # ---- f32 random with cold items ----
    pr = make_problem(200, 3000, 24, f32, mean_c=60, seed=15)
    Bc = pr["B"].copy()
    Bc[rng.random(3000) < 0.15] = 0
    variants = [dict(k=10, cumulative=False, noise=True, seed=2 ** 40 + 17), dict(k=10, cumulative=True, noise=True, seed=3)]
    run_case("g9_noise_cold_f32", pr["A"], Bc, pr["train"], pr["test"], variants, f32)


if __name__ == "__main__":
    main()
