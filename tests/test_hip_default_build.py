"""GPU: the HIP path against the build of the reference its users actually run.

SURVEY.md section 8(c), contract item (4).  The parity contract is pinned to the CANONICAL build of the reference (no -march:
`dot1` is the strict k-ordered fma chain, src/recometrics.hpp:99-112) -- bit for bit.  The reference's default build is
`-march=native` (setup.py:35-41): its `dot1` (src/recometrics.hpp:84-112) is vectorised and reassociated, so its scores differ
from the canonical ones in the last bits, and a candidate within rounding of a positive's score may change places with it.
oracle/_ref/librecometrics_ref_fast.so is that build at the portable -march=x86-64-v3 (AVX2 + FMA).  Contract: on non-dyadic
data every metric of every user agrees within 1e-5, except users with a near tie -- a positive whose neighbour in the ranking
lies within 2^-20 relative (plus what reassociating k products can move) -- and every user beyond 1e-5 must be such a user
(oracle/ties.py compare_with_default_build).  The record printed is the one bench.py carries as `parity_vs_default_build`."""
import os

import numpy as np
import pytest

from test_hip_parity import hip, hip_calc  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
NT = max(1, min(256, os.cpu_count() or 1))


def _compare(hip, A, B, train, test, K, dtype=np.float32):
    from oracle.oracle import Reference, reference_available
    from oracle.ties import compare_with_default_build
    if not reference_available(fast=True):
        pytest.skip("oracle/_ref/librecometrics_ref_fast.so is not on this box")
    got = hip_calc(hip, A, B, train, test, K, dtype=dtype)
    want = Reference(fast=True).calc(A, B, train, test, K, nthreads=NT, dtype=dtype)
    amax = float(np.abs(A).max() * np.abs(B).max())
    rec = compare_with_default_build(got, want, lambda who: hip.debug_scores(np.ascontiguousarray(A[who]), B), train, test, K, dtype,
                                     A.shape[1], amax)
    print("parity_vs_default_build:", {k: v for k, v in rec.items() if k != "rule"})
    assert rec["nan_mismatch"] == 0, rec
    assert rec["ok"], "users beyond 1e-5 of the reference's vectorised build WITHOUT a near tie: %s" % rec
    return rec


def test_default_build_c1(hip):
    """BASELINE C1: 1,000 users x 5,000 items x 64 factors, K = 10, every user"""
    from recometrics_amd.synth import CONFIGS, make_problem
    m, n, k, dtype, K, mean_c, seed = CONFIGS["C1"]
    pr = make_problem(m, n, k, dtype, mean_c=mean_c, seed=seed)
    rec = _compare(hip, pr["A"], pr["B"], pr["train"], pr["test"], K)
    assert rec["users"] == m and rec["max_abs_diff_unexplained"] <= 1e-5


def test_default_build_c2_sample(hip):
    """BASELINE C2 at its item count: a stratified sample of 2,048 of the 138,493 users (heaviest rows, streamed users, cold and
    skipped users, first and last block)"""
    import bench
    from recometrics_amd.synth import CONFIGS
    m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
    host = bench.host_problem(m, n, k, mean_c, seed, dtype)
    users = bench.stratified_users(host, 2048)
    A, B, tr, te = bench.sub_problem(host, users)
    rec = _compare(hip, np.ascontiguousarray(A), B, tr, te, K)
    assert rec["users"] == users.shape[0]


def test_default_build_north_star_192(hip):
    """the north-star shape: 192 users x 1,000,000 items x 128 factors (item splits, ~0.5 exact ties per user among 1M fp32 scores)"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(192, 1_000_000, 128, np.float32, mean_c=100, seed=100)
    _compare(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10)


def test_default_build_f64(hip):
    """fp64 (the reference's double `dot1`, vectorised in the default build as well): 300 users x 20,000 items x 96 factors"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(300, 20_000, 96, np.float64, mean_c=80, seed=8)
    rec = _compare(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=np.float64)
    assert rec["max_abs_diff_unexplained"] <= 1e-9
