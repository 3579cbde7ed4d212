"""GPU parity tests: the HIP path (through the C-ABI) against the oracle and the reference's golden vectors.

Bar (BASELINE.json north_star): scores, top-K index lists, hit counts and positive ranks BIT-EXACT;
fp32/fp64 metric values within 1e-5 of the CPU reference (they are in fact bit-identical except where the
reference forms ROC-AUC in x87 long double).
"""
import os

import numpy as np
import pytest

from _util import assert_close, assert_same_bits, golden_cases, load_golden
from _util import same_bits as _util_same_bits
from oracle.ties import tie_pairs_per_user

pytestmark = pytest.mark.gpu
TOL = 1e-5
NT = max(1, min(64, os.cpu_count() or 1))      # host threads for the oracle (the GPU box has many cores)


@pytest.fixture(scope="module")
def hip():
    from recometrics_amd import _binding
    _binding.load()
    assert _binding.device_count() > 0, "no HIP device visible"
    return _binding


def hip_calc(hip, A, B, train, test, k, metrics=("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr"),
             cumulative=False, noise=False, cold=True, min_items_pool=2, min_pos_test=1, seed=1, dtype=np.float32, **_):
    from oracle.oracle import NAMES
    A = np.ascontiguousarray(A, dtype=dtype)
    B = np.ascontiguousarray(B, dtype=dtype)
    trp, tri = [np.ascontiguousarray(x, dtype=np.int32) for x in train[:2]]
    tep, tei = [np.ascontiguousarray(x, dtype=np.int32) for x in test[:2]]
    tev = np.ascontiguousarray(test[2], dtype=dtype) if len(test) > 2 and test[2] is not None else np.ones(tei.shape[0], dtype)
    want = {name: (name in metrics) for name in hip.METRIC_ORDER}
    outs = hip.calc_metrics(A, A.shape[1], B, B.shape[1], trp, tri, tep, tei, tev, k, want, cumulative, noise, cold,
                            min_items_pool, min_pos_test, 1, seed)
    return {NAMES[name]: arr for name, arr in zip(hip.METRIC_ORDER, outs) if want[name]}


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 8, 24, 33, 41, 50, 64, 80, 96, 100, 128, 129, 200, 256, 300, 512, 520, 1000, 1024, 1500])
def test_mfma_scores_are_the_k_ordered_fma_chain(hip, oracle, k):
    """The sweep's v_mfma_f32_32x32x2_f32 contraction == strict index-order fmaf chain (reference dot1), bit for bit."""
    rng = np.random.default_rng(k)
    A = rng.standard_normal((70, k)).astype(np.float32)
    B = rng.standard_normal((333, k)).astype(np.float32)
    got = hip.debug_scores(A, B)
    want = oracle.scores(A, B)
    assert_same_bits(got, want, "scores k=%d" % k)


@pytest.mark.parametrize("k", [3, 8, 40, 64, 100, 130, 256, 300, 512, 513, 700, 1024])
def test_mfma_f64_scores_are_the_k_ordered_fma_chain(hip, oracle, k):
    """v_mfma_f64_16x16x4_f64 contraction == strict index-order fma chain (reference dot1, double), bit for bit."""
    rng = np.random.default_rng(100 + k)
    A = rng.standard_normal((37, k)) * np.exp(rng.uniform(-20, 20, (37, 1)))
    B = rng.standard_normal((150, k))
    got = hip.debug_scores(A, B)
    want = oracle.scores(A, B, dtype=np.float64)
    assert_same_bits(got, want, "f64 scores k=%d" % k)


def test_mfma_scores_subnormal_and_special_values(hip, oracle):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((40, 32)).astype(np.float32)
    B = rng.standard_normal((130, 32)).astype(np.float32)
    A[3] *= 1e-30; B[7] *= 1e-12            # subnormal products / sums
    A[5, 4] = np.inf; B[9, 2] = -np.inf; B[11, 0] = np.nan
    got = hip.debug_scores(A, B)
    want = oracle.scores(A, B)
    assert_same_bits(got, want, "special values")


# ---------------------------------------------------------------------------------------------------------------------
def _skip_user_with_nan_test_value(case, name, got, want, inp):
    """NaN test VALUES: the reference feeds them to std::partial_sort (unspecified); the device path declares the
    user's NDCG NaN.  Exclude exactly those users from the NDCG comparison."""
    if name != "NDCG@K":
        return got, want
    tep, tev = inp["test"][0], inp["test"][2]
    bad = np.array([np.isnan(tev[tep[u]:tep[u + 1]]).any() for u in range(len(tep) - 1)])
    if bad.any():
        got, want = got.copy(), want.copy()
        got[bad] = 0; want[bad] = 0
    return got, want


@pytest.mark.parametrize("case", [c for c in golden_cases()])
def test_golden_fixtures(hip, case):
    dtype, inp, variants = load_golden(case)
    for vi, (kw, expected) in enumerate(variants):
        got = hip_calc(hip, inp["A"], inp["B"], inp["train"], inp["test"], dtype=dtype, **kw)
        assert set(got) == set(expected)
        for name in expected:
            g, w = _skip_user_with_nan_test_value(case, name, got[name], expected[name], inp)
            assert_close(g, w, TOL, "%s v%d %s %s" % (case, vi, kw, name))
            # break_ties_with_noise=True: the reference's mt19937(seed + user) noise is reproduced bit for bit, so what it
            # decides -- top-K membership and order, the rank of every test item -- is the reference's (ROC-AUC apart:
            # x87 long double there)
            if kw.get("noise") and name != "ROC_AUC" and case.startswith(("g9_", "g6_", "g4_")):
                assert_same_bits(g, w, "%s v%d %s %s (bitwise, noise on)" % (case, vi, kw, name))


_REF = []


def _reference():
    """the real reference as a second checker, when its compiled library is present (it is on the GPU box: oracle/_ref travels)"""
    if not _REF:
        from oracle.oracle import Reference, reference_available
        _REF.append(Reference() if reference_available() else None)
    return _REF[0]


def test_the_compiled_reference_is_the_second_checker():
    """oracle/_ref/librecometrics_ref.so is built in the build container and travels with the snapshot; without it the
    tests below would silently fall back to the restatement alone"""
    assert _reference() is not None, "oracle/_ref/librecometrics_ref.so is missing on this box"


# ---------------------------------------------------------------------------------------------------------------------
def _check_against_oracle(hip, oracle, pr, k, dtype=np.float32, **kw):
    want_rank = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], k, dtype=dtype, nthreads=NT)
    trp, tri = pr["train"]
    tep, tei = pr["test"][:2]
    got_rank = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, k)
    assert (got_rank["status"] == want_rank["status"]).all()
    assert (got_rank["topk_idx"] == want_rank["topk_idx"]).all(), "top-K index lists differ"
    assert_same_bits(got_rank["topk_score"], want_rank["topk_score"], "top-K scores")
    assert (got_rank["pos_rank"] == want_rank["pos_rank"]).all(), "positive ranks differ"
    ref = _reference()
    for cumulative in (False, True):
        want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], k, cumulative=cumulative, dtype=dtype, nthreads=NT, **kw)
        got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], k, cumulative=cumulative, dtype=dtype, **kw)
        if ref is not None:
            # not only the restatement: the REAL reference (oracle/_ref, compiled from /root/reference by oracle/Makefile; the
            # library travels to the GPU box) on the same inputs.  Noise off: exact ties are ordered by item id here and by
            # libstdc++'s sort there (deviation D4) -- the synthetic factors have none.
            real = ref.calc(pr["A"], pr["B"], pr["train"], pr["test"], k, cumulative=cumulative, dtype=dtype, nthreads=NT, **kw)
            # Noise off: scores that are EXACTLY equal (with 40,000 fp32 scores per user a handful of pairs always are) are ordered
            # by item id here and in the restatement, and by libstdc++'s introsort in the reference -- deviation D4.  Such a pair
            # changes a metric only when one of the two is a test item of the user (oracle/ties.py), so: EVERY user that differs
            # from the compiled reference -- in the last bit of any metric, or by more than 1e-5 in ROC-AUC (formed in x87 long
            # double there: an ulp apart for everybody) -- must have a candidate whose score equals one of its positives' scores
            # exactly, or the test fails.  A tied pair moves a positive by one rank, 1 / (positives x negatives) of ROC-AUC.
            n_users = len(tep) - 1
            differing = np.zeros(n_users, bool)
            for name in real:
                assert (np.isnan(got[name]) == np.isnan(real[name])).all(), name
                if name != "ROC_AUC":
                    assert_close(got[name], real[name], TOL, "%s cumulative=%s vs the compiled reference" % (name, cumulative))
                    diff = ~_util_same_bits(got[name], real[name]).reshape(n_users, -1).all(axis=1)
                    if name == "PR_AUC" and os.environ.get("RM_STREAM_BUDGET_MB") == "0":
                        diff &= np.diff(tep) <= 63          # deviation D7 (fallback only): per-chunk partial sums, a few ulp(fp64), checked below at 1e-12
                    differing |= diff
            d_roc = np.zeros(n_users)
            if "ROC_AUC" in real:
                r64 = np.nan_to_num(real["ROC_AUC"].astype(np.float64))
                d_roc = np.abs(np.nan_to_num(got["ROC_AUC"].astype(np.float64)) - r64) / np.maximum(1.0, np.abs(r64))   # (relative beyond magnitude 1, see assert_close)
                differing |= d_roc > TOL
            pairs = np.zeros(n_users, np.int64)
            if differing.any():
                who = np.flatnonzero(differing)
                sc = hip.debug_scores(np.ascontiguousarray(pr["A"][who], dtype), np.ascontiguousarray(pr["B"], dtype))
                noisy = bool(kw.get("noise"))
                pairs[who] = tie_pairs_per_user(sc, pr["train"], pr["test"], who, noise_zone=(2.0 ** -14 if noisy and dtype == np.float32 else None))
                unexplained = who[pairs[who] == 0]
                assert unexplained.size == 0, "users %s differ from the compiled reference without an exact tie on a positive" % unexplained[:8].tolist()
            npos_u = np.diff(tep).astype(np.float64)
            nneg_u = np.maximum(pr["B"].shape[0] - np.diff(trp) - npos_u, 1)
            assert (d_roc <= TOL + pairs / np.maximum(npos_u * nneg_u, 1)).all(), "ROC_AUC vs the compiled reference: %g" % d_roc.max()
        # PR_AUC of a user with more than 63 test items is assembled from per-chunk partial sums (finalize):
        # same terms, different association than the reference's single running sum -> a few ulp(fp64), checked at 1e-12
        # (only when such users take one sweep slot per chunk, RM_STREAM_BUDGET_MB=0; by default their ranks come from
        # their stored score rows and the sum is the reference's own left-to-right one: bitwise for every user)
        one_chunk = (np.diff(tep) <= 63) if os.environ.get("RM_STREAM_BUDGET_MB") == "0" else np.ones(len(tep) - 1, bool)
        for name in want:
            assert_close(got[name], want[name], TOL, "%s cumulative=%s" % (name, cumulative))
            if name == "PR_AUC":
                assert_same_bits(got[name][one_chunk], want[name][one_chunk], "%s single-chunk users (bitwise)" % name)
                assert_close(got[name][~one_chunk], want[name][~one_chunk], 1e-12, "%s multi-chunk users" % name)
            elif name != "ROC_AUC":
                assert_same_bits(got[name], want[name], "%s cumulative=%s (bitwise)" % (name, cumulative))


@pytest.mark.parametrize("m,n,k,K,mean_c", [
    (1000, 5000, 64, 10, 50),      # BASELINE config C1
    (300, 2000, 32, 7, 40),
    (257, 1111, 50, 5, 30),        # ragged: users % 32, items % 64, factors % 8 all non-zero
    (64, 40000, 128, 10, 100),     # few users, long item axis: item splits + merge of partial lists
    (500, 3000, 16, 40, 60),       # larger K (top-K lists out of LDS)
    (200, 50000, 128, 100, 10),    # BASELINE config C4's factor count and K: HBM append buffers + wave compaction
    (300, 9000, 100, 256, 30),     # largest supported K for HBM lists
    (150, 4000, 200, 10, 40),      # > 128 factors: the factor axis is streamed in chunks of 128
    (90, 3000, 500, 5, 30),
    (200, 5000, 128, 20, 120),     # BASELINE config C3's factor count and K: replace-the-minimum lists in HBM + pending buffers
    (160, 7000, 128, 32, 120),     # largest K of that scheme
    (900, 4000, 128, 20, 90),      # enough users for a depth split: shallow blocks with LDS lists beside deep ones with HBM lists
    (100, 6000, 256, 20, 100),     # the same with a streamed factor axis (prefetched user factors)
    (120, 3000, 600, 10, 40),      # more than 512 factors: chunk count of the factor axis at run time
    (70, 2500, 1024, 30, 60),
    (150, 4000, 48, 300, 60),      # k_metrics > 256: every user streamed, top-K picked from the stored rows (k_select_topk)
    (90, 2500, 130, 1000, 40),
    (40, 700, 16, 699, 30),        # k_metrics = n - 1
])
def test_random_problem_vs_oracle(hip, oracle, m, n, k, K, mean_c):
    from recometrics_amd.synth import make_problem
    pr = make_problem(m, n, k, np.float32, mean_c=mean_c, seed=m + n)
    _check_against_oracle(hip, oracle, pr, K)


@pytest.mark.parametrize("m,n,k,K,mean_c", [
    (200, 3000, 19, 10, 30),
    (150, 5000, 64, 5, 50),
    (70, 20000, 256, 50, 50),      # BASELINE config C5's factor count and K (fp64, lists out of LDS, item splits)
    (129, 1027, 100, 7, 40),
    (60, 2000, 400, 5, 30),        # > 256 factors in fp64
    (80, 6000, 256, 30, 120),      # fp64 replace-the-minimum lists in HBM (too large for LDS, K <= 32) + pending buffers
    (96, 9000, 64, 12, 60),        # fp64 LDS lists + pending buffers
    (500, 3000, 128, 28, 90),      # fp64 depth split: shallow blocks with LDS lists beside deep ones with HBM lists
    (60, 2000, 600, 10, 40),       # fp64, more than 512 factors
    (100, 3000, 40, 400, 50),      # fp64, k_metrics > 256
    (90, 2600, 24, 1000, 40),      # fp64, k_collect_topk's largest sort (2,048 entries of 64-bit keys in registers)
    (90, 2600, 24, 1300, 40),
])
def test_random_problem_vs_oracle_f64(hip, oracle, m, n, k, K, mean_c):
    from recometrics_amd.synth import make_problem
    pr = make_problem(m, n, k, np.float64, mean_c=mean_c, seed=m + n + 1)
    _check_against_oracle(hip, oracle, pr, K, dtype=np.float64)


@pytest.mark.parametrize("k", [17, 24, 33, 40, 48, 50, 56, 72, 80, 96, 100, 104])
@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_HBM_LISTS": "1"}, {"RM_DEBUG_NSUB2": "1"}])
def test_every_factor_group_count_has_its_kernel(hip, oracle, k, env, monkeypatch):
    """factor counts between the powers of two run on kernels of their own group count -- 3, 5, 6, 7 groups of 8 factors up to 64
    factors (three or two sub-tiles, LDS or HBM lists), 10, 12, 13 up to 128 -- instead of being padded to the next power of two
    (50 factors, the reference notebook's model: 56, not 64; 100: 104, not 128): the whole pipeline against the oracle, K = 10 and
    K = 40 (append-buffer lists)"""
    from recometrics_amd.synth import make_problem
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    pr = make_problem(300, 5000 + k, k, np.float32, mean_c=50, seed=1000 + k)
    _check_against_oracle(hip, oracle, pr, 10)
    if not env:
        _check_against_oracle(hip, oracle, pr, 40)
    if k <= 64 and k % 16 and not env.get("RM_DEBUG_NSUB2"):         # fp64 has the group counts up to 8 as well
        pr64 = make_problem(200, 4000 + k, k, np.float64, mean_c=50, seed=2000 + k)
        _check_against_oracle(hip, oracle, pr64, 10 if not env else 40, dtype=np.float64)


@pytest.mark.parametrize("env", [{"RM_DEBUG_NO_PENDING": "1"}, {"RM_DEBUG_HBM_LISTS": "1"}, {"RM_DEBUG_EXT_TOPK": "1"},
                                 {"RM_DEBUG_NO_TRAIN_BITS": "1"}, {"RM_DEBUG_NO_SEED": "1"}, {"RM_DEBUG_SPLITS": "5"},
                                 {"RM_DEBUG_SPLITS": "2,1,5"}, {"RM_DEBUG_SPLITS": "1,2,3"}, {"RM_DEBUG_SPLITS": "3,1,7", "RM_DEBUG_HBM_LISTS": "1"},
                                 {"RM_DEBUG_HBM_LISTS": "1", "RM_DEBUG_NO_PENDING": "1"},
                                 {"RM_DEBUG_NSUB2": "1"}, {"RM_DEBUG_NSUB2": "1", "RM_DEBUG_NO_PENDING": "1"}, {"RM_DEBUG_NO_POS_FLAT": "1"}])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_list_schemes_agree(hip, oracle, env, dtype, monkeypatch):
    """the same problem through every top-K list scheme the sweep has (the library reads the switches at load; conftest.py makes it read them again after every monkeypatch.setenv), with the
    CSR cursor instead of the dense train rows, without the seeded bounds, with another item split count, with a two-level
    grid ("S,tail user blocks,tail splits": the cheapest user blocks cut into more item ranges than the others):
    LDS lists without pending buffers, HBM replace-the-minimum lists with and without them, and (fp32, <= 64 factors,
    where three item sub-tiles per step are the default) the two-sub-tile block"""
    from recometrics_amd.synth import make_problem
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    if dtype == np.float64 and any(key in env for key in ("RM_DEBUG_NO_PENDING", "RM_DEBUG_HBM_LISTS", "RM_DEBUG_NSUB2")):
        monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1000000")      # (fp64 takes the lane buffers for every k_metrics since round 6: these are the lists' switches)
    pr = make_problem(150, 9000, 48, dtype, mean_c=60, seed=5)
    _check_against_oracle(hip, oracle, pr, 12, dtype=dtype)


@pytest.mark.parametrize("splits", [None, "2,2,3", "1,1,4"])
def test_depth_split_launches_with_a_two_level_grid(hip, oracle, splits, monkeypatch):
    """128 factors, K = 32, users with deep and with shallow positive trees: the LDS lists fit next to the shallow blocks'
    tables only, so the sweep runs as two launches side by side (deep blocks with HBM lists, shallow ones with LDS lists);
    with a two-level grid the tail user blocks are the shallow launch's first, then the deep one's.  (Since round 6 a k_metrics of
    21 and more takes the lane buffers, and up to 20 the lists fit LDS at every depth: the depth split is what remains when the lane
    buffers do not fit the device's memory -- here they are switched off.)"""
    from recometrics_amd.synth import make_problem
    monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1000000")
    if splits:
        monkeypatch.setenv("RM_DEBUG_SPLITS", splits)
    pr = make_problem(700, 6000, 128, np.float32, mean_c=120, seed=77)
    _check_against_oracle(hip, oracle, pr, 32)
    assert hip.timings()["sweep_launches"] == 2, hip.timings()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("budget", ["0", "1", None])
def test_long_test_rows_streamed_or_chunked(hip, oracle, dtype, budget, monkeypatch):
    """users with more than 63 test items: streamed through HBM (default), one sweep slot per chunk of 63 when the
    score rows do not fit the budget (forced here: 0 = never stream, 1 MB = too small for these rows)"""
    from recometrics_amd.synth import make_problem
    if budget is not None:
        monkeypatch.setenv("RM_STREAM_BUDGET_MB", budget)
    pr = make_problem(700, 7000, 40, dtype, mean_c=260, seed=31)         # about half of the users have long rows
    if budget == "1":
        monkeypatch.setenv("RM_STREAM_BUDGET_MB", "0")                       # the bitwise rule of the checker: chunked sums
        _check_against_oracle(hip, oracle, pr, 10, dtype=dtype)
        monkeypatch.setenv("RM_STREAM_BUDGET_MB", "1")
        got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype)
        monkeypatch.setenv("RM_STREAM_BUDGET_MB", "0")
        ref = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype)
        for name in ref:
            assert_same_bits(got[name], ref[name], "budget too small == never stream: " + name)
    else:
        _check_against_oracle(hip, oracle, pr, 10, dtype=dtype)


@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_NO_TRAIN_BITS": "1"}, {"RM_DEBUG_NO_POS_FLAT": "1"}])
@pytest.mark.parametrize("dtype,k", [(np.float32, 50), (np.float32, 3), (np.float64, 17)])
def test_scores_of_the_test_entries_by_entry(hip, oracle, dtype, k, env, monkeypatch):
    """the positives' scores by ENTRY (k_pos_scores_flat: 64 consecutive test entries per wavefront, whatever rows they belong to) and
    the kernels that tell it about its entries (k_entry_users; the train-item test from the dense rows' kernel, or k_test_masked
    without them -- RM_DEBUG_NO_TRAIN_BITS, fp64): empty test rows between the others, rows of one entry, rows that straddle the
    64-entry pieces, users that are not evaluated (no train items), factor counts that leave ragged 16-byte vectors and ragged
    pieces, one user whose train row is longer than k_test_masked's LDS piece (searched in memory), test items that are train
    items at the start, inside and at the end of rows.  Against the oracle, bit for bit; RM_DEBUG_NO_POS_FLAT: the by-slot kernels."""
    from recometrics_amd.synth import make_factors
    rng = np.random.default_rng(4242)
    m, n = 333, 30011
    A, B = make_factors(m, n, k, dtype, seed=9)
    lens = [0, 1, 1, 63, 64, 65, 2, 0, 0, 127, 5, 62, 1, 200, 0, 31, 33, 700, 1, 1]
    rows_tr, rows_te = [], []
    for u in range(m):
        nte = lens[u % len(lens)]
        ntr = 0 if u % 17 == 5 else (11000 if u == 40 else int(rng.integers(1, 400)))
        items = rng.permutation(n)[: nte + ntr]
        te, tr = np.sort(items[:nte]), np.sort(items[nte:])
        if nte and u % 3 == 0:                                     # some test items are train items too: first, middle, last of the row
            tr = np.union1d(tr, te[[0, nte // 2, nte - 1]])
        rows_te.append(te); rows_tr.append(tr)
    def csr(rows):
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        return indptr, np.concatenate(rows).astype(np.int32)
    tep, tei = csr(rows_te)
    pr = {"A": A, "B": B, "train": csr(rows_tr), "test": (tep, tei, rng.integers(1, 21, size=tei.shape[0]).astype(dtype))}
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    _check_against_oracle(hip, oracle, pr, 7, dtype=dtype)
    _check_against_oracle(hip, oracle, pr, 7, dtype=dtype, cold=False)                # (users without train items are not evaluated)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("K", [65, 100, 256])
def test_ideal_dcg_with_k_metrics_beyond_the_finalize_buffer(hip, oracle, dtype, K):
    """NDCG's ideal DCG needs the min(K, positives) largest test VALUES in order.  k_finalize sorts up to 64 of them in its own
    buffer; with k_metrics > 64 every row of more than 64 test items goes through k_top_values (a wavefront per user, its list read
    back by k_finalize) instead of a repeated selection on one thread.  Rows of 1 ... 600 test items around every threshold (64, K,
    256), values with ties, zeros and negatives (the reference stops the ideal sum at the first non-positive value when the K-th
    is negative, src/recometrics.hpp:868-961), single and cumulative outputs -- against the oracle, bit for bit."""
    from recometrics_amd.synth import make_factors
    rng = np.random.default_rng(77 + K)
    lens = [1, 40, 63, 64, 65, 66, 99, 100, 101, 128, 200, 255, 256, 257, 300, 600]
    m, n, k = 64, 6000, 16
    A, B = make_factors(m, n, k, dtype, seed=21)
    rows_tr, rows_te, vals = [], [], []
    for u in range(m):
        nte = lens[u % len(lens)]
        items = rng.permutation(n)[: nte + 50]
        rows_te.append(np.sort(items[:nte])); rows_tr.append(np.sort(items[nte:]))
        kind = u // len(lens)
        v = rng.integers(-3, 6, size=nte) if kind % 2 == 0 else rng.integers(1, 4, size=nte)       # with / without non-positive values
        if kind == 3 and nte > 2: v[:] = 2                                                         # every value tied
        vals.append(v.astype(dtype))
    def csr(rows):
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        return indptr, np.concatenate(rows).astype(np.int32)
    tep, tei = csr(rows_te)
    pr = {"A": A, "B": B, "train": csr(rows_tr), "test": (tep, tei, np.concatenate(vals).astype(dtype))}
    _check_against_oracle(hip, oracle, pr, K, dtype=dtype)


def test_second_plan_keeps_the_users_the_tie_noise_flagged(hip, monkeypatch):
    """fp32 tie noise + a budget too small for the streamed users' score rows: the plan is made twice (second time with the long rows
    in chunks), while the positives' scores -- whose kernel flags the users with a test item the noise can move, and counts them --
    were launched beside the FIRST plan and are not made again: the count survives the second plan's reset (a lost count would skip
    the exact pass and leave the un-noised values).  Cold items (all-zero factors) make sure users are flagged.  Bit for bit the
    call that never streams."""
    from recometrics_amd.synth import make_problem
    pr = make_problem(500, 7000, 40, np.float32, mean_c=260, seed=33)
    rng = np.random.default_rng(8)
    pr["B"] = pr["B"].copy()
    pr["B"][rng.random(7000) < 0.2] = 0
    monkeypatch.setenv("RM_STREAM_BUDGET_MB", "0")
    ref = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, noise=True, seed=7)
    quiet = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, noise=False)
    assert any((np.asarray(ref[name]) != np.asarray(quiet[name]))[~np.isnan(np.asarray(quiet[name]))].any() for name in ref), "the noise changed nothing: the case does not test it"
    monkeypatch.setenv("RM_STREAM_BUDGET_MB", "1")
    got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, noise=True, seed=7)
    for name in ref:
        assert_same_bits(got[name], ref[name], "second plan, tie noise: " + name)


@pytest.mark.parametrize("noise", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_test_items_that_are_also_train_items(hip, oracle, dtype, noise):
    """rows whose TEST entries partly (or all) repeat TRAIN entries: the reference never scores a train item (src/recometrics.hpp:491-497),
    so such a test item is no candidate and can never be hit, while it still counts in the row lengths (:479-482, recall's
    denominator); users with few and with many (streamed) test items, one whose test row lies entirely inside its train row"""
    from recometrics_amd.synth import make_problem
    rng = np.random.default_rng(4242)
    pr = make_problem(260, 3000 + 11, 40, dtype, mean_c=220, seed=88)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    rows = []
    for u in range(trp.shape[0] - 1):
        tr = tri[trp[u]:trp[u + 1]]; te = tei[tep[u]:tep[u + 1]]
        if u % 3 == 0: add = te[rng.random(te.shape[0]) < 0.3]            # a third of its test items also in the train row
        elif u % 11 == 1: add = te                                        # every test item is a train item
        else: add = te[:0]
        rows.append(np.union1d(tr, add).astype(np.int32))
    trp2 = np.concatenate([[0], np.cumsum([r.shape[0] for r in rows])]).astype(np.int32)
    pr["train"] = (trp2, np.concatenate(rows).astype(np.int32))
    _check_against_oracle(hip, oracle, pr, 10, dtype=dtype, noise=noise, seed=5)


@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_RANK_GENERIC": "1"}, {"RM_DEBUG_NO_SIDE": "1"},
                                 {"RM_DEBUG_RANK_GENERIC": "1", "RM_DEBUG_NO_SIDE": "1"}, {"RM_DEBUG_NO_FUSED_AUC": "1"},
                                 {"RM_DEBUG_NO_DEFER_AUC": "1"}, {"RM_DEBUG_NO_TEST_MASK": "1"}, {"RM_DEBUG_NO_POS_FLAT": "1"},
                                 {"RM_DEBUG_NO_POS_FLAT": "1", "RM_DEBUG_NO_SIDE": "1"}, {"RM_DEBUG_NO_POS_BESIDE": "1"}])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_streamed_ranks_fast_routine_and_side_stream(hip, oracle, env, dtype, monkeypatch):
    """streamed users of every table depth the fast routine of k_rank_streamed is built for (64 .. 1023 test items: 128, 256, 512
    and 1024 table entries) next to shorter and longer rows, an item count that leaves a ragged end for the generic routine, and a
    masked stretch (a user whose train row covers a run of items): the ranks with the fast routine / the generic one, on the side
    stream / on the main one, are the oracle's"""
    from recometrics_amd.synth import make_factors
    rng = np.random.default_rng(909)
    m, n, k = 96, 9000 + 37, 24
    A, B = make_factors(m, n, k, dtype, seed=3)
    # (2048 and more: the table no longer fits LDS, the search and the counters are in global memory and k_auc_streamed walks them --
    # next to rows whose block of k_rank_streamed walks its own counts)
    lens = [3, 40, 63, 64, 65, 100, 127, 128, 200, 255, 256, 400, 511, 512, 800, 1023, 1024, 1500, 2047, 2048, 2600, 4000]
    rows_tr, rows_te = [], []
    for u in range(m):
        nte = lens[u % len(lens)]
        items = rng.permutation(n)[: nte + 150]
        te = np.sort(items[:nte]); tr = np.sort(items[nte:])
        if u % 7 == 0:                                              # a run of consecutive train items: whole masked stretches in the row
            tr = np.union1d(tr, np.setdiff1d(np.arange(2000, 2600), te))
        rows_te.append(te); rows_tr.append(tr)
    def csr(rows):
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        return indptr, np.concatenate(rows).astype(np.int32)
    tep, tei = csr(rows_te)
    pr = {"A": A, "B": B, "train": csr(rows_tr), "test": (tep, tei, rng.integers(1, 21, size=tei.shape[0]).astype(dtype))}
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    _check_against_oracle(hip, oracle, pr, 10, dtype=dtype)


@pytest.mark.parametrize("dtype,budget", [(np.float32, None), (np.float64, None), (np.float32, "1"), (np.float64, "1")])
def test_tie_noise_rankings_equal_the_oracle(hip, oracle, dtype, budget, monkeypatch):
    """break_ties_with_noise=True (the API default): ordered top-K lists, the rank of every test item and all metrics equal
    the oracle's, which draws from std::mt19937 / std::uniform_real_distribution themselves (and is pinned on the
    reference's own noise-on outputs, tests/golden/g9_noise_*).  15 % cold items (all-zero factors): blocks of exactly tied
    scores that only the noise orders.  budget = 1 MB: the noise rows are made in several batches."""
    from recometrics_amd.synth import make_problem
    if budget is not None:
        monkeypatch.setenv("RM_NOISE_BUDGET_MB", budget)
    pr = make_problem(300, 5000, 20, dtype, mean_c=80, seed=41)
    rng = np.random.default_rng(5)
    pr["B"] = pr["B"].copy()
    pr["B"][rng.random(5000) < 0.15] = 0
    seed = 2 ** 33 + 5
    want_rank = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype, nthreads=NT, noise=True, seed=seed)
    trp, tri = pr["train"]
    tep, tei = pr["test"][:2]
    got_rank = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, 10,
                        break_ties_with_noise=True, seed=seed)
    assert (got_rank["status"] == want_rank["status"]).all()
    assert (got_rank["topk_idx"] == want_rank["topk_idx"]).all(), "top-K index lists differ"
    assert_same_bits(got_rank["topk_score"], want_rank["topk_score"], "top-K scores (with their noise)")
    assert (got_rank["pos_rank"] == want_rank["pos_rank"]).all(), "positive ranks differ"
    for cumulative in (False, True):
        want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], 10, cumulative=cumulative, dtype=dtype, nthreads=NT, noise=True, seed=seed)
        got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, cumulative=cumulative, dtype=dtype, noise=True, seed=seed)
        for name in want:
            assert_close(got[name], want[name], TOL, name)
            if name != "ROC_AUC":
                assert_same_bits(got[name], want[name], "%s cumulative=%s (bitwise, noise on)" % (name, cumulative))
        ref = _reference()
        if ref is not None:
            # the compiled reference with ITS noise on the same inputs: in fp64 the noise separates every pair of scores, so
            # everything but the x87 ROC-AUC is bit for bit; in fp32 scores of ordinary magnitude can stay exactly tied (D4)
            real = ref.calc(pr["A"], pr["B"], pr["train"], pr["test"], 10, cumulative=cumulative, dtype=dtype, nthreads=NT, noise=True, seed=seed)
            for name in real:
                assert_close(got[name], real[name], TOL, name + " vs the compiled reference, noise on")
                if name != "ROC_AUC" and dtype == np.float64:
                    assert_same_bits(got[name], real[name], "%s cumulative=%s vs the compiled reference (bitwise, noise on)" % (name, cumulative))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tie_noise_values_bit_for_bit(hip, oracle, dtype):
    """every candidate's noisy score (the whole ranking of a few users: k_metrics = candidates - 1) equals the oracle's bit
    for bit.  Pins the libstdc++ arithmetic of the noise itself -- r * (b - a) rounded BEFORE a is added: a fused multiply-add
    there (the compiler's default contraction) leaves one value in ten an ulp off, which no top-10 list notices but an exact
    tie of two noisy scores does (found by scratch/fuzz.py)."""
    from recometrics_amd.synth import make_problem
    pr = make_problem(5, 3000, 12, dtype, mean_c=200, seed=77)
    pr["B"] = pr["B"].copy()
    pr["B"][np.random.default_rng(1).random(3000) < 0.3] = 0           # zero scores: the noise IS the score
    trp, tri = pr["train"]
    tep, tei = pr["test"][:2]
    K = 3000 - int(np.diff(trp).max()) - 1
    for seed in (1, 2 ** 40 + 9):
        want = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, nthreads=NT, noise=True, seed=seed)
        got = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, K,
                       break_ties_with_noise=True, seed=seed)
        assert (got["topk_idx"] == want["topk_idx"]).all(), "full rankings differ"
        assert_same_bits(got["topk_score"], want["topk_score"], "noisy scores of every candidate")
        assert (got["pos_rank"] == want["pos_rank"]).all()


def test_rank_outputs_do_not_depend_on_the_previous_call(hip, oracle):
    """the device workspace is reused between calls: users that are never ranked (here: all candidates fit into K, no
    AUC tables) must report rank 0 whatever an earlier, larger problem left behind"""
    from recometrics_amd.synth import make_problem
    big = make_problem(1200, 4000, 16, np.float32, mean_c=60, seed=21)
    _check_against_oracle(hip, oracle, big, 10)
    small = make_problem(700, 20, 1, np.float32, mean_c=3.3, seed=22)
    _check_against_oracle(hip, oracle, small, 19)


def test_random_shapes_and_options(hip, oracle):
    """a fixed-seed sample of scratch/fuzz.py: random shapes, precisions and option combinations against the oracle"""
    from recometrics_amd.synth import make_problem
    rng = np.random.default_rng(2024)
    for _ in range(60):
        dtype = np.float32 if rng.random() < 0.65 else np.float64
        m = int(rng.choice([1, 7, 33, 100, 300, 700]))
        n = int(rng.choice([20, 97, 300, 1111, 4000]))
        k = int(rng.choice([1, 5, 16, 33, 64, 100, 128, 200]))
        K = int(min(n - 1, rng.choice([1, 3, 10, 20, 33, 60])))
        mean_c = float(min(n / 6, rng.choice([4, 20, 60, 150, 500])))
        kw = dict(cold=bool(rng.random() < 0.7), min_items_pool=int(rng.choice([1, 2, 10])), min_pos_test=int(rng.choice([1, 1, 3])))
        pr = make_problem(m, n, k, dtype, mean_c=mean_c, seed=int(rng.integers(1 << 30)))
        for cumulative in (False, True):
            want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], K, cumulative=cumulative, dtype=dtype, nthreads=NT, **kw)
            got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], K, cumulative=cumulative, dtype=dtype, **kw)
            for name in want:
                assert_close(got[name], want[name], TOL, "%s %s m=%d n=%d k=%d K=%d %s" % (name, dtype.__name__, m, n, k, K, kw))


def test_heavy_users_many_positives(hip, oracle):
    """users with far more than 63 positives (multi-slot AUC chunks) and with dense train rows"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(96, 6000, 24, np.float32, mean_c=700, seed=77)
    _check_against_oracle(hip, oracle, pr, 10)


def test_edge_users_block(hip, oracle):
    """no test items, cold users, k_leq_n and only-NDCG users inside one call (fixture g4 shapes, fresh data)"""
    _, inp, _ = load_golden("g4_edge_users_f32")
    for K in (2, 4, 10):
        for cold in (True, False):
            for cumulative in (False, True):
                want = oracle.calc(inp["A"], inp["B"], inp["train"], inp["test"], K, cumulative=cumulative, cold=cold)
                got = hip_calc(hip, inp["A"], inp["B"], inp["train"], inp["test"], K, cumulative=cumulative, cold=cold)
                for name in want:
                    assert_close(got[name], want[name], TOL, "%s K=%d cold=%s cum=%s" % (name, K, cold, cumulative))


def test_metric_subsets(hip, oracle):
    from recometrics_amd.synth import make_problem
    pr = make_problem(200, 1500, 20, np.float32, mean_c=30, seed=9)
    for metrics in (("p",), ("ndcg",), ("ap", "rr"), ("roc",), ("roc", "pr"), ("hit", "rr"), ("p", "ndcg", "pr")):
        want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], 6, metrics=metrics)
        got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 6, metrics=metrics)
        assert set(got) == set(want)
        for name in want:
            assert_close(got[name], want[name], TOL, "%s of %s" % (name, metrics))


def test_python_api_drop_in(hip, oracle):
    """calc_reco_metrics: reference signature, DataFrame naming, dtype rule, item_biases folding"""
    import pandas as pd
    from scipy.sparse import csr_array
    from recometrics_amd import calc_reco_metrics
    from recometrics_amd.synth import make_problem
    pr = make_problem(120, 900, 16, np.float32, mean_c=25, seed=21)
    m, n = 120, 900
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    Xtr = csr_array((np.ones(tri.shape[0], np.float32), tri, trp), shape=(m, n))
    Xte = csr_array((tev, tei, tep), shape=(m, n))
    df = calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=3, all_metrics=True, break_ties_with_noise=False)
    assert isinstance(df, pd.DataFrame) and df.shape[0] == m
    assert list(df.columns) == ["P@3", "TP@3", "R@3", "AP@3", "TAP@3", "NDCG@3", "Hit@3", "RR@3", "ROC_AUC", "PR_AUC"]
    assert all(dt == np.float32 for dt in df.dtypes)
    want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], 3)
    for col, key in zip(df.columns, want):
        assert_close(df[col].to_numpy(), want[key], TOL, col)
    d = calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=4, as_df=False, cumulative=True, break_ties_with_noise=False)
    assert d["K"] == 4 and set(d) == {"P@K", "AP@K", "NDCG@K", "K"} and d["P@K"].shape == (m, 4)
    bias = np.linspace(-1, 1, n).astype(np.float32)
    d2 = calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=4, item_biases=bias, as_df=False, break_ties_with_noise=False)
    Ab = np.c_[pr["A"], np.ones((m, 1), np.float32)]; Bb = np.c_[pr["B"], bias.reshape(-1, 1)]
    wantb = oracle.calc(Ab, Bb, pr["train"], pr["test"], 4, metrics=("p", "ap", "ndcg"))
    for key in ("P@K", "AP@K", "NDCG@K"):
        assert_close(d2[key], wantb[key], TOL, "biases " + key)
    with pytest.raises(ValueError):
        calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=n + 1)


def test_c_abi_error_statuses(hip):
    """status codes instead of exceptions (include/recometrics_hip.h): unsupported shapes and bad arguments come back
    as errors with a message, never as wrong numbers; the next valid call works"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(40, 300, 8, np.float32, mean_c=20, seed=3)
    good = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 5)
    with pytest.raises(ValueError, match="leading dimension"):
        hip.calc_metrics(pr["A"], 4, pr["B"], 8, pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], pr["test"][2], 5,
                         {name: True for name in hip.METRIC_ORDER}, False, False, True, 2, 1, 1, 1)
    with pytest.raises(ValueError, match="k_metrics"):
        hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 0)
    again = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 5)
    for name in good:
        assert_same_bits(again[name], good[name], "repeat after an error: " + name)


def test_python_api_modes(hip, oracle):
    """float64 inputs, strided factor matrices (lda > k), X_train=None, the non-personalised mode (A = B = None with
    item biases), cumulative DataFrame output -- reference recometrics/__init__.py:429-436,:469-473,:560-562,:615-626"""
    from scipy.sparse import csr_array
    from recometrics_amd import calc_reco_metrics
    from recometrics_amd.synth import make_problem
    m, n, k = 90, 700, 12
    pr = make_problem(m, n, k, np.float64, mean_c=25, seed=33)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    Xtr = csr_array((np.ones(tri.shape[0]), tri, trp), shape=(m, n))
    Xte = csr_array((tev, tei, tep), shape=(m, n))
    # float64 + strided A (a column slice of a wider row-major array keeps lda = 20)
    wide = np.zeros((m, 20)); wide[:, :k] = pr["A"]
    d = calc_reco_metrics(Xtr, Xte, wide[:, :k], pr["B"], k=5, all_metrics=True, as_df=False, break_ties_with_noise=False)
    want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], 5, dtype=np.float64)
    assert d["P@K"].dtype == np.float64
    for key in want:
        assert_close(d[key], want[key], TOL, "f64 strided " + key)
    # mixed dtypes => float64 (only float32 when BOTH are float32)
    d = calc_reco_metrics(Xtr, Xte, pr["A"].astype(np.float32), pr["B"], k=5, as_df=False, break_ties_with_noise=False)
    assert d["P@K"].dtype == np.float64
    # no train matrix
    d = calc_reco_metrics(None, Xte, pr["A"], pr["B"], k=5, ndcg=False, roc_auc=True, as_df=False, break_ties_with_noise=False)
    empty = (np.zeros(m + 1, np.int32), np.zeros(0, np.int32))
    want = oracle.calc(pr["A"], pr["B"], empty, pr["test"], 5, metrics=("p", "ap", "roc"), dtype=np.float64)
    for key in want:
        assert_close(d[key], want[key], TOL, "no train " + key)
    # non-personalised: scores = item biases
    bias = np.random.default_rng(1).standard_normal(n)
    d = calc_reco_metrics(Xtr, Xte, None, None, k=5, item_biases=bias, as_df=False, break_ties_with_noise=False)
    want = oracle.calc(np.ones((m, 1)), bias.reshape(-1, 1), pr["train"], pr["test"], 5, metrics=("p", "ap", "ndcg"), dtype=np.float64)
    for key in want:
        assert_close(d[key], want[key], TOL, "biases only " + key)
    # cumulative DataFrame with AUC columns (the reference raises IndexError here; documented fix)
    df = calc_reco_metrics(Xtr, Xte, pr["A"], pr["B"], k=3, precision=True, average_precision=False, ndcg=False,
                           roc_auc=True, cumulative=True, break_ties_with_noise=False)
    assert list(df.columns) == ["P@1", "P@2", "P@3", "ROC_AUC"] and df.shape == (m, 4)


def test_north_star_shape_vs_oracle(hip, oracle):
    """n = 1M items x 128 factors (the north-star shape), 192 users: item splits + shared thresholds + exact ties at
    scale (fp32 scores collide with positives' scores ~0.5 times per user here)."""
    from recometrics_amd.synth import make_problem
    pr = make_problem(192, 1_000_000, 128, np.float32, mean_c=100, seed=100)
    _check_against_oracle(hip, oracle, pr, 10)


def test_full_size_properties(hip, oracle):
    """BASELINE-scale shape (C2-like slice): size-independent properties, plus the oracle on a sub-sample of the users
    (users are independent, so rows [0, 384) of the big run must equal the oracle run on those rows alone)"""
    from recometrics_amd.sharding import slice_csr
    from recometrics_amd.synth import make_problem
    pr = make_problem(4096, 26744, 64, np.float32, mean_c=144, seed=102)
    single = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10)
    sub_tr = slice_csr(pr["train"][0], pr["train"][1], None, 0, 384)[:2]
    sub_te = slice_csr(pr["test"][0], pr["test"][1], pr["test"][2], 0, 384)
    want = oracle.calc(pr["A"][:384], pr["B"], sub_tr, sub_te, 10, nthreads=NT)
    for name in want:
        assert_close(single[name][:384], want[name], TOL, "sub-sample " + name)
    cum = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, cumulative=True)
    for name in ("P@K", "TP@K", "R@K", "AP@K", "TAP@K", "Hit@K", "RR@K"):
        assert_same_bits(cum[name][:, -1], single[name], name + ": cumulative column K == single-K")
    roc = single["ROC_AUC"]
    assert np.nanmin(roc) >= 0 and np.nanmax(roc) <= 1 and abs(np.nanmean(roc) - 0.5) < 0.02
    assert (np.diff(cum["R@K"], axis=1) >= 0).all() and (np.diff(cum["Hit@K"], axis=1) >= 0).all()
    rk = hip.rank(pr["A"], pr["B"], pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], 10)
    sc = rk["topk_score"]
    assert (np.diff(sc, axis=1) <= 0).all(), "top-K scores must be sorted descending"
    # top-K never contains a train item
    trp, tri = pr["train"]
    for u in range(0, 4096, 97):
        assert not set(rk["topk_idx"][u]).intersection(tri[trp[u]:trp[u + 1]])


@pytest.mark.parametrize("noise", [False, True])
def test_baseline_c2_at_its_full_user_count(hip, noise):
    """BASELINE C2 as quoted -- all 138,493 users x 26,744 items x 64 factors, K = 10, all ten metrics -- through the
    host-pointer entry: the m-driven machinery (a grid of 1,256 sweep blocks in several rounds with a two-level tail, 26 k
    streamed users = 2.8 GB of score rows addressed beyond 2^31 bytes, user batches) at full size, checked against the
    REAL reference (oracle/_ref) on a stratified sample of 2,560 users: heaviest test rows, streamed users, the deepest
    LDS tables, cold and skipped users, the first and last user blocks, a random remainder.  noise=True is the API default
    (the reference's mt19937(seed + user) stream: it sees the whole workload with the other users' test rows emptied, so the
    sampled users keep their indices)."""
    import bench
    from oracle.oracle import NAMES, Oracle, Reference, reference_available
    from recometrics_amd.synth import CONFIGS, make_factors, make_interactions
    m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
    _, B = make_factors(1, n, k, dtype, seed)
    A, _ = make_factors(m, 1, k, dtype, seed + 1)
    trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed)
    host = dict(A=A, B=B, train=(trp, tri), test=(tep, tei, tev))
    want_flags = {name: True for name in hip.METRIC_ORDER}
    outs = hip.calc_metrics(A, k, B, k, trp, tri, tep, tei, tev, K, want_flags, False, noise, True, 2, 1, 1, 77)
    users = bench.stratified_users(host, 2560)
    assert (np.diff(tep)[users] > 63).sum() >= 300 and users[-1] == m - 1
    sA, sB, str_, ste = bench.sub_problem(host, users)
    impl = Reference() if reference_available() else Oracle()
    if noise:
        # the reference seeds a user's noise by its ORIGINAL index: it gets the whole workload with the other users' test rows emptied
        # (it skips a user without test items at once, src/recometrics.hpp:439-448)
        fA, fB, ftr, fte = bench.masked_problem(host, users)
        want = impl.calc(fA, fB, ftr, fte, K, nthreads=min(256, os.cpu_count() or 1), noise=True, seed=77, dtype=dtype)
        want = {name: arr[users] for name, arr in want.items()}
    else:
        want = impl.calc(sA, sB, str_, ste, K, nthreads=min(256, os.cpu_count() or 1), noise=noise, seed=77, dtype=dtype)
    differing = np.zeros(users.shape[0], bool)
    for name, arr in zip(hip.METRIC_ORDER, outs):
        assert_close(arr[users], want[NAMES[name]], TOL, "C2 full size, noise=%s: %s" % (noise, name))
        if name != "roc":
            differing |= ~_util_same_bits(arr[users], want[NAMES[name]])
    # exactly tied scores (a few pairs among 26,744 fp32 scores per user; the fp32 noise of 1e-12 does not separate scores of
    # ordinary magnitude) are ordered by item id here and by libstdc++'s sort there (deviation D4): every user that differs in
    # the last bit of a metric must have a candidate whose score equals one of its positives' exactly (oracle/ties.py)
    if differing.any() and isinstance(impl, Reference):
        who = np.flatnonzero(differing)
        sc = hip.debug_scores(sA[who], sB)
        pairs = tie_pairs_per_user(sc, str_, ste, who, noise_zone=(2.0 ** -14 if noise else None))
        assert (pairs > 0).all(), "users %s differ from the compiled reference without an exact tie on a positive" % users[who[pairs == 0]][:8].tolist()
    else:
        assert not differing.any()
    roc = outs[hip.METRIC_ORDER.index("roc")]
    assert abs(np.nanmean(roc) - 0.5) < 0.005 and np.isnan(roc).sum() == (np.diff(tep) == 0).sum()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE shapes at their full ITEM counts (user-sliced: users are independent, so a slice of the users exercises
# exactly the n-driven machinery -- item splits and the merge of partial lists, > 2^32-byte offsets into the packed
# item image, append buffers compacted across splits, histogram counts in the millions -- at a size the oracle
# finishes in seconds).  Reference src/recometrics.hpp:501-512 switches to size_t addressing for exactly these sizes.
def _big_problem(m, n, k, dtype, mean_c, seed, heavy=()):
    """make_problem with the item factors drawn in row chunks (bounded host memory at n = 10M) and, optionally, the
    test rows of the first len(heavy) users replaced by rows of the given lengths (users with > 63 / > 256 positives)."""
    from recometrics_amd.synth import make_interactions
    rng = np.random.default_rng(seed)
    scale = dtype(1.0 / np.sqrt(k))
    A = (rng.standard_normal((m, k), dtype=np.float32)).astype(dtype) * scale
    B = np.empty((n, k), dtype=dtype)
    step = 1 << 20
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        B[r0:r1] = rng.standard_normal((r1 - r0, k), dtype=np.float32).astype(dtype) * scale
    trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed)
    if heavy:
        rows_i = [tei[tep[u]:tep[u + 1]] for u in range(m)]
        rows_v = [tev[tep[u]:tep[u + 1]] for u in range(m)]
        for u, cnt in enumerate(heavy):
            taken = set(tri[trp[u]:trp[u + 1]].tolist())
            pick = np.array(sorted(set(rng.integers(0, n, size=cnt + 64).tolist()) - taken)[:cnt], dtype=np.int32)
            rows_i[u] = pick
            rows_v[u] = rng.integers(1, 21, size=pick.shape[0]).astype(dtype)
        tep = np.concatenate([[0], np.cumsum([r.shape[0] for r in rows_i])]).astype(np.int32)
        tei = np.concatenate(rows_i).astype(np.int32)
        tev = np.concatenate(rows_v).astype(dtype)
    return {"A": A, "B": B, "train": (trp, tri), "test": (tep, tei, tev)}


def test_baseline_c3_item_count(hip, oracle):
    """C3: 380,000 items x 128 factors fp32, K = 20 single and cumulative (HBM replace-the-minimum lists, depth split,
    item splits), 640 users incl. two with more than 63 test items"""
    pr = _big_problem(640, 380_000, 128, np.float32, 48, 103, heavy=(100, 300))
    _check_against_oracle(hip, oracle, pr, 20)


def test_baseline_c4_item_count(hip, oracle):
    """C4: 10,000,000 items x 128 factors fp32, K = 100, all metrics (append buffers + compaction, a 5.12 GB packed item
    image, histogram counts in the millions), 64 users incl. one with > 63 and one with > 256 test items"""
    pr = _big_problem(64, 10_000_000, 128, np.float32, 10, 104, heavy=(90, 300))
    _check_against_oracle(hip, oracle, pr, 100)


def test_baseline_c5_item_count(hip, oracle):
    """C5: 500,000 items x 256 factors fp64, K = 50 (streamed factor axis, fp64 append buffers), 96 users"""
    pr = _big_problem(96, 500_000, 256, np.float64, 50, 105, heavy=(70,))
    _check_against_oracle(hip, oracle, pr, 50, dtype=np.float64)


_DENSE_ROWS_SCRIPT = r"""
import json, os, sys, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch                                           # (before the library: the order bench.py loads them in)
import bench
from recometrics_amd import _binding as hip
from _util import same_bits
torch.cuda.set_device(0); hip.load(); hip.set_device(0)
m, n = 360_000, 100_000
prob = bench.DeviceProblem(torch, torch.device("cuda", 0), m, n, 16, 20, 4242, 10)
stream = torch.cuda.current_stream().cuda_stream
def run(env):
    os.environ.update(env)
    hip.reload_switches()                                  # (the library reads its switches at load: read them again)
    assert hip.load().rm_release_workspace() == 0
    free0 = torch.cuda.mem_get_info(0)[0]
    out = torch.empty_like(prob.out)
    prob.step(hip, stream, out)
    torch.cuda.synchronize()
    used = free0 + out.numel() * out.element_size() - torch.cuda.mem_get_info(0)[0]
    for key in env:
        del os.environ[key]
    hip.reload_switches()
    return out, used
cursor, used_cursor = run({"RM_DEBUG_NO_TRAIN_BITS": "1"})
dense, used_dense = run({})
res = bench.parity_check(prob, dense, 600, binding=hip)
print(json.dumps({"used_dense": used_dense, "used_cursor": used_cursor, "rows_bytes": m * ((n + 191) // 192 * 6) * 4,
                  "same_bits": bool(same_bits(dense.cpu().numpy(), cursor.cpu().numpy()).all()), "parity": res}))
"""


def test_dense_train_rows_of_many_users_in_one_call(hip):
    """360,000 users x 100,000 items x 16 factors through the device entry in ONE call: the dense train rows of the fp32 sweep are
    4.5 GB -- beyond the 1 GiB they always get (rm_lib.hip dense_rows_fit: a quarter of the free HBM, 8 GiB at most), row offsets
    beyond 2^32 bytes -- checked against the compiled reference on a stratified sample that holds the first and the last user
    block, and bit for bit against the same call over the CSR cursor (RM_DEBUG_NO_TRAIN_BITS).  The workspace grows by the rows.
    (A process of its own: the inputs live in torch tensors, and torch wants to be loaded before the library.)"""
    import json
    import subprocess
    import sys
    from oracle.oracle import reference_available
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the score rows of streamed users get a fixed budget in both runs, so that the two workspaces differ by the train rows alone)
    env = dict(os.environ, RM_STREAM_BUDGET_MB="4096")
    res = subprocess.run([sys.executable, "-c", _DENSE_ROWS_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert got["rows_bytes"] > 2 ** 32 and got["used_dense"] - got["used_cursor"] > 0.9 * got["rows_bytes"], got
    assert got["same_bits"], "dense train rows and the CSR cursor disagree"
    assert got["parity"]["ok"] and (got["parity"]["checker"] == "reference") == reference_available(), got


# ---------------------------------------------------------------------------------------------------------------------
# boundary behaviour (reference src/recometrics.hpp:359-436, recometrics/wrapper.pyx:226-323)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_zero_users_is_a_no_op(hip, dtype):
    """m == 0: the reference's loop over users does not run and the call returns normally (:428-437)"""
    A = np.zeros((0, 8), dtype)
    B = np.ones((50, 8), dtype)
    zp = np.zeros(1, np.int32)
    outs = hip.calc_metrics(A, 8, B, 8, zp, np.zeros(0, np.int32), zp, np.zeros(0, np.int32), np.zeros(0, dtype), 5,
                            {name: True for name in hip.METRIC_ORDER}, False, True, True, 2, 1, 1, 1)
    assert all(o.size == 0 for o in outs)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_empty_test_matrix_with_null_indices(hip, oracle, dtype):
    """no test entry at all: the bindings hand over NULL for the empty index / value arrays; every user is NaN"""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((40, 8)).astype(dtype)
    B = rng.standard_normal((300, 8)).astype(dtype)
    trp = np.arange(0, 41 * 3, 3, dtype=np.int32)
    tri = np.tile(np.array([1, 5, 9], np.int32), 40)
    tep = np.zeros(41, np.int32)
    want = {name: name != "ndcg" for name in hip.METRIC_ORDER}
    outs = hip.calc_metrics(A, 8, B, 8, trp, tri, tep, np.zeros(0, np.int32), np.zeros(0, dtype), 5, want, False, False, True, 2, 1, 1, 1)
    for name, o in zip(hip.METRIC_ORDER, outs):
        if want[name]:
            assert np.isnan(o).all(), name
    ref = oracle.calc(A, B, (trp, tri), (tep, np.zeros(0, np.int32), np.zeros(0, dtype)), 5, dtype=dtype,
                      metrics=tuple(nm for nm in hip.METRIC_ORDER if nm != "ndcg"))
    assert all(np.isnan(v).all() for v in ref.values())
    from recometrics_amd import build as rb
    import importlib.util, os as _os
    if _os.path.exists(rb.cython_module_path()):
        from recometrics_amd import _cy
        outs2 = _cy.calc_metrics(A, 8, B, 8, trp, tri, tep, np.zeros(0, np.int32), np.zeros(0, dtype), 5, want, False, False, True, 2, 1, 1, 1)
        for name, o in zip(hip.METRIC_ORDER, outs2):
            if want[name]:
                assert np.isnan(o).all(), "cython binding: " + name


def test_ndcg_without_test_values_is_an_error(hip):
    """deviation D8: the reference dereferences the null pointer (:870-874); here RM_ERR_INVALID"""
    A = np.ones((4, 4), np.float32)
    B = np.ones((30, 4), np.float32)
    p = np.arange(5, dtype=np.int32)
    i = np.arange(4, dtype=np.int32)
    want = {name: name == "ndcg" for name in hip.METRIC_ORDER}
    with pytest.raises(ValueError):
        hip.calc_metrics(A, 4, B, 4, np.zeros(5, np.int32), np.zeros(0, np.int32), p, i, np.zeros(0, np.float32), 3, want, False, False, True, 2, 1, 1, 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_repeated_test_columns_fill_the_top_k(hip, oracle, dtype):
    """a non-canonical test row (an item listed twice; the reference only sorts the rows): the seeded K-th-best bound must count
    candidates, not entries -- the ordered top-K lists stay complete and equal the oracle's"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(200, 3000, 8, dtype, mean_c=60, seed=9)
    tep, tei, tev = pr["test"]
    rows_i, rows_v, newp = [], [], [0]
    for u in range(200):
        it = tei[tep[u]:tep[u + 1]]
        rep = np.sort(np.concatenate([it, it, it[:3]]))               # every test item two or three times
        rows_i.append(rep); rows_v.append(np.ones(rep.shape[0], dtype)); newp.append(newp[-1] + rep.shape[0])
    test = (np.array(newp, np.int32), np.concatenate(rows_i).astype(np.int32), np.concatenate(rows_v))
    trp, tri = pr["train"]
    for K in (5, 12):
        want = oracle.rank(pr["A"], pr["B"], pr["train"], test, K, dtype=dtype, nthreads=NT)
        got = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, test[0], test[1], K)
        assert (got["status"] == want["status"]).all() and (got["status"] == 0).sum() > 50
        assert (got["topk_idx"][got["status"] == 0] >= 0).all(), "incomplete top-K list"
        assert (got["topk_idx"] == want["topk_idx"]).all()
        assert_same_bits(got["topk_score"], want["topk_score"], "top-K scores")


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE C3 and C5 at their full USER counts, one device call each (the m-driven machinery -- grids of thousands of sweep blocks
# in many rounds, the two-level tail, plan scans over a million users, metric blocks of hundreds of MB -- at the size the configs
# are quoted on; the item-count side is covered above).  Inputs live in torch tensors: a process of its own.
_FULL_M_SCRIPT = r"""
import json, os, sys, time, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch
import bench
from recometrics_amd import _binding as hip
from recometrics_amd.synth import CONFIGS, make_interactions_fast
from _util import same_bits
torch.cuda.set_device(0); hip.load(); hip.set_device(0)
dev = torch.device("cuda", 0)
name, cumulative = %(name)r, %(cumulative)r
m, n, k, dtype, K, mean_c, seed = CONFIGS[name]
t0 = time.time()
rng = np.random.default_rng(seed)
sc = np.float32(1.0 / np.sqrt(k))
def factors(rows):
    X = np.empty((rows, k), dtype=dtype)
    for r0 in range(0, rows, 1 << 18):
        r1 = min(rows, r0 + (1 << 18))
        X[r0:r1] = rng.standard_normal((r1 - r0, k), dtype=np.float32) * sc
    return X
A, B = factors(m), factors(n)
trp, tri, tep, tei, tev = make_interactions_fast(m, n, mean_c, dtype, seed)
host = dict(A=A, B=B, train=(trp, tri), test=(tep, tei, tev))
t_gen = time.time() - t0
prob = bench.DeviceProblem(torch, dev, m, n, k, mean_c, seed, K, dtype, cumulative=cumulative, host=host)
stream = torch.cuda.current_stream().cuda_stream
prob.step(hip, stream); torch.cuda.synchronize()
t1 = time.time(); prob.step(hip, stream); torch.cuda.synchronize(); t_call = time.time() - t1
tm = hip.timings()
res = {"users": m, "gen_s": t_gen, "call_ms": t_call * 1e3, "sweep_ms": tm["sweep_ms"], "sweep_blocks": tm["sweep_blocks"], "item_splits": tm["item_splits"],
       "mfma_frac": 2.0 * n * k * m / (tm["sweep_ms"] * 1e-3) / 1e12 / (157.3 if dtype == np.float32 else 78.6)}
# properties that do not depend on the size
ntest = np.diff(tep)
roc = prob.metric(prob.out, 8).cpu().numpy()
res["roc_mean"] = float(np.nanmean(roc)); res["roc_nan"] = int(np.isnan(roc).sum()); res["users_without_test"] = int((ntest == 0).sum())
res["roc_in_range"] = bool(np.nanmin(roc) >= 0 and np.nanmax(roc) <= 1)
p_at = prob.metric(prob.out, 0).cpu().numpy(); r_at = prob.metric(prob.out, 2).cpu().numpy(); hit = prob.metric(prob.out, 6).cpu().numpy()
if cumulative:
    res["recall_monotone"] = bool((np.diff(r_at, axis=1) >= 0).all()); res["hit_monotone"] = bool((np.diff(hit, axis=1) >= 0).all())
    pk, hk = p_at[:, -1], hit[:, -1]
else:
    res["recall_monotone"] = res["hit_monotone"] = True
    pk, hk = p_at, hit
res["hit_is_p_positive"] = bool(((pk > 0) == (hk > 0)).all())
pkk = np.nan_to_num(pk.astype(np.float64) * K)
res["p_grid"] = bool(np.allclose(pkk, np.rint(pkk), atol=1e-4))                  # P@K = hits / K
# users are independent: the first 512 users of the big call == a call over those 512 users alone, bit for bit
sub = 512
hs = dict(A=A[:sub], B=B, train=(trp[:sub + 1], tri[:max(int(trp[sub]), 1)]), test=(tep[:sub + 1], tei[:int(tep[sub])], tev[:int(tep[sub])]))
small = bench.DeviceProblem(torch, dev, sub, n, k, mean_c, seed, K, dtype, cumulative=cumulative, host=hs)
small.step(hip, stream); torch.cuda.synchronize()
ok = True
for i in range(10):
    ok &= bool(same_bits(prob.metric(prob.out, i)[:sub].cpu().numpy(), small.metric(small.out, i).cpu().numpy()).all())
res["first_users_equal_their_own_call"] = ok
# a stratified sample (heaviest rows, streamed users, cold users, first and last block) against the compiled reference
res["parity"] = bench.parity_check(prob, prob.out, %(sample)d, cpu_seconds=%(cpu_s)f, binding=hip)
print(json.dumps(res))
"""


def _full_m_run(name, cumulative, sample, cpu_s):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _FULL_M_SCRIPT % {"root": root, "name": name, "cumulative": cumulative, "sample": sample, "cpu_s": cpu_s}],
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    got = json.loads(res.stdout.strip().splitlines()[-1])
    print(name, "at its full user count:", {k: v for k, v in got.items() if k != "parity"}, got["parity"])
    return got


def _full_m_asserts(got, m):
    from oracle.oracle import reference_available
    assert got["users"] == m
    assert got["roc_in_range"] and abs(got["roc_mean"] - 0.5) < 0.005 and got["roc_nan"] == got["users_without_test"], got
    assert got["recall_monotone"] and got["hit_monotone"] and got["hit_is_p_positive"] and got["p_grid"], got
    assert got["first_users_equal_their_own_call"], "the first users of the big call differ from a call over them alone"
    assert got["parity"]["ok"] and (got["parity"]["checker"] == "reference") == reference_available(), got["parity"]


def test_baseline_c3_at_its_full_user_count(hip):
    """C3 as BASELINE quotes it before sharding: 1,000,000 users x 380,000 items x 128 factors fp32, cumulative K = 1..20, all ten
    metrics, ONE device call (7,813 user blocks; 162 values per user = 648 MB of outputs)"""
    got = _full_m_run("C3", True, 600, 8.0)
    _full_m_asserts(got, 1_000_000)


def test_baseline_c5_at_its_full_user_count(hip):
    """C5: 200,000 users x 500,000 items x 256 factors fp64, K = 50, all ten metrics, ONE device call"""
    got = _full_m_run("C5", False, 160, 8.0)
    _full_m_asserts(got, 200_000)


def test_baseline_c4_at_its_full_user_count(hip):
    """C4: 100,000 users x 10,000,000 items x 128 factors fp32, K = 100 + ROC / PR-AUC, ONE device call (782 user blocks over a
    5.12 GB item matrix; the lane buffers and k_collect_topk of k_metrics = 100 at full size; ~2.5 s of device work)"""
    got = _full_m_run("C4", False, 48, 12.0)
    _full_m_asserts(got, 100_000)


# ---------------------------------------------------------------------------------------------------------------------
# k_metrics beyond the LDS lists (round 6): per-lane append buffers + lane-parallel selection in the sweep, k_collect_topk behind it
@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_LANE_CAP_MIN": "1"}, {"RM_DEBUG_LANE_CAP_MIN": "1", "RM_DEBUG_SPLITS": "3,2,5"},
                                 {"RM_DEBUG_LANE_CAP_MIN": "1", "RM_DEBUG_NO_SEED": "1", "RM_DEBUG_NO_TRAIN_BITS": "1"}])
@pytest.mark.parametrize("dtype,K", [(np.float32, 33), (np.float32, 100), (np.float32, 256), (np.float64, 40), (np.float64, 50), (np.float64, 256)])
def test_lane_buffers_and_selections(hip, oracle, dtype, K, env, monkeypatch):
    """the smallest lane buffers that work force a selection every few tiles (the default size sees two per item range at these sizes);
    several item ranges per user make the shared bound overtake a lane's own entries (lane_select's `below` case) and give
    k_collect_topk many sources"""
    from recometrics_amd.synth import make_problem
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    pr = make_problem(140, 12000, 40 if dtype == np.float32 else 24, dtype, mean_c=70, seed=600 + K)
    _check_against_oracle(hip, oracle, pr, K, dtype=dtype)


@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_NO_TRAIN_BITS": "1"}, {"RM_DEBUG_SPLITS": "3,2,5", "RM_DEBUG_LANE_CAP_MIN": "1"}, {"RM_DEBUG_NO_SEED": "1"}])
@pytest.mark.parametrize("sample,K,mean_c", [(64, 21, 20), (64, 60, 300), (256, 33, 70), (256, 200, 70), (1024, 100, 70), (1024, 500, 900), (2048, 256, 70)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sample_seeds_of_the_lane_buffers(hip, oracle, sample, K, mean_c, env, dtype, monkeypatch):
    """the lane buffers' bounds start at the K-th best candidate of the first `sample` items (k_seed_from_sample over the sweep's own
    scores of them): forced at every sample size, with K close to and beyond what the sample holds once the train items are out
    (no seed then), with the train items masked by the dense rows and by the kernel's own walk of the sparse rows"""
    from recometrics_amd.synth import make_problem
    monkeypatch.setenv("RM_DEBUG_SAMPLE_SEED", str(sample))
    monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1")
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    pr = make_problem(150, 6000, 40 if dtype == np.float32 else 24, dtype, mean_c=mean_c, seed=4000 + sample + K)
    _check_against_oracle(hip, oracle, pr, K, dtype=dtype)


@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_HBM_LISTS": "1"}, {"RM_DEBUG_NSUB2": "1", "RM_DEBUG_NO_TRAIN_BITS": "1"}, {"RM_DEBUG_SPLITS": "3,2,5"}])
@pytest.mark.parametrize("sample,K,k,m,mean_c", [(64, 5, 40, 150, 20), (64, 12, 64, 150, 70), (256, 18, 16, 150, 70), (256, 13, 128, 150, 70), (256, 30, 40, 150, 900),
                                                  (64, 20, 128, 900, 90)])
def test_sample_seeds_of_the_lists(hip, oracle, sample, K, k, m, mean_c, env, monkeypatch):
    """the replace-the-minimum lists start from the sample's bound as well (by default from k_metrics = 12 with 256 items): forced here at
    every k_metrics, with three sub-tiles per step (the sample's DUMP launch then has an item image of its own and walks the sparse
    train rows), two, lists in HBM, several item ranges, the depth split's two launches (900 users, 128 factors)"""
    from recometrics_amd.synth import make_problem
    monkeypatch.setenv("RM_DEBUG_SAMPLE_SEED", str(sample))
    monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1000000")
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    pr = make_problem(m, 5000, k, np.float32, mean_c=mean_c, seed=5000 + sample + K)
    _check_against_oracle(hip, oracle, pr, K, dtype=np.float32)


@pytest.mark.parametrize("env", [{}, {"RM_DEBUG_LANE_CAP_MIN": "1"}])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lane_selection_cuts_exact_ties_by_item(hip, oracle, dtype, env, monkeypatch):
    """item factors repeated in runs: every score occurs 40 times in a row, far more exact ties at a user's K-th best than a
    selection's slack -- the bisection ends on the exact score and the ties are cut by item id (rm_list.hpp lane_select)"""
    from recometrics_amd.synth import make_problem
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    pr = make_problem(70, 8000, 16, dtype, mean_c=40, seed=77)
    B = pr["B"].copy()
    B[:] = B[(np.arange(B.shape[0]) // 40) * 40]
    pr["B"] = B
    trp, tri = pr["train"]
    tep, tei = pr["test"][:2]
    want = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], 48, dtype=dtype, nthreads=NT)
    got = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(B, dtype), trp, tri, tep, tei, 48)
    assert (got["status"] == want["status"]).all()
    assert (got["topk_idx"] == want["topk_idx"]).all(), "top-K index lists differ"
    assert_same_bits(got["topk_score"], want["topk_score"], "top-K scores")
    assert (got["pos_rank"] == want["pos_rank"]).all(), "positive ranks differ"


@pytest.mark.parametrize("dtype,k,K,noise", [(np.float64, 300, 60, True), (np.float64, 200, 33, True), (np.float64, 300, 60, False),
                                             (np.float32, 40, 33, False), (np.float32, 24, 100, True), (np.float32, 300, 60, True)])
def test_lane_buffers_in_the_kernels_that_spill_the_buffer_base(hip, oracle, dtype, k, K, noise, monkeypatch):
    """the sweep kernels whose register pressure spills the lane buffers' base pointer (fp64 with the factor axis in 128-factor chunks
    and the run-time switches -- tie noise, RM_DEBUG_NO_SPEC --, fp32 at 24-40 factors) restore it with v_readlane right in front of the
    appends' inline asm: without the wait states inside the statement the stores went to a stale base (memory faults; round 6)"""
    from recometrics_amd.synth import make_problem
    if not noise:
        monkeypatch.setenv("RM_DEBUG_NO_SPEC", "1")
    pr = make_problem(200, 9000, k, dtype, mean_c=80, seed=900 + k)
    _check_against_oracle(hip, oracle, pr, K, dtype=dtype, noise=noise)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("K", [1, 3, 8, 21])
def test_small_lane_buffers_never_run_into_the_next_waves(hip, oracle, dtype, K, monkeypatch):
    """every k_metrics through the lane buffers (RM_DEBUG_LANE_MIN_K): with fewer than 60 entries per lane the staggered first selection
    of the second sub-tile (3/4 of the buffer) lay beyond the level that keeps a tile's appends inside the buffer -- lanes overflowed
    into the next wave's rows and other users lost candidates (found by scratch/fuzz_r6.sh)"""
    from recometrics_amd.synth import make_problem
    monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1")
    for seed in (1234, 1235):
        pr = make_problem(1500, 300, 100 if dtype == np.float32 else 40, dtype, mean_c=50, seed=seed)
        _check_against_oracle(hip, oracle, pr, K, dtype=dtype)


@pytest.mark.parametrize("m,n,k,K,mean_c", [
    (96, 9000, 64, 12, 60),        # fp64 LDS lists + pending buffers
    (80, 6000, 256, 30, 120),      # fp64 replace-the-minimum lists in HBM (too large for LDS) + pending buffers
    (500, 3000, 128, 28, 90),      # fp64 depth split: shallow blocks with LDS lists beside deep ones with HBM lists
    (70, 20000, 256, 50, 50),      # k_metrics = 50 on the HBM lists
])
def test_f64_lists_when_the_lane_buffers_are_off(hip, oracle, m, n, k, K, mean_c, monkeypatch):
    """fp64 takes the lane buffers for every k_metrics since round 6; the replace-the-minimum lists (LDS, HBM, the depth split) remain
    for lane buffers that do not fit the device's memory -- here with the lane buffers switched off"""
    from recometrics_amd.synth import make_problem
    monkeypatch.setenv("RM_DEBUG_LANE_MIN_K", "1000000")
    pr = make_problem(m, n, k, np.float64, mean_c=mean_c, seed=m + n + 1)
    _check_against_oracle(hip, oracle, pr, K, dtype=np.float64)
