"""Test infrastructure (imported by tests/ and bench.py's parity leg only): which users have EXACTLY tied scores that can
change a metric.

The reference orders equal scores by whatever libstdc++'s introsort leaves (src/recometrics.hpp:537-563, unspecified); the
device and the restatement order them by item id (DESIGN.md deviation D4).  A tie changes a metric only when one of the tied
candidates is a test item of the user: two tied negatives swap without a trace, inside the top-K list or across its boundary.
So a user whose metrics differ from the compiled reference must have a candidate whose score equals the score of one of its
positives -- anything else is a bug, and the parity tests fail on it."""
import numpy as np


def tie_pairs_per_user(scores, train, test, users=None, noise_zone=None):
    """scores[i] = dense score row of user users[i] (bit-exact scores, e.g. rm_debug_scores_*); returns for every such user the
    number of (positive, other candidate) pairs with exactly equal scores.  Train items are no candidates
    (src/recometrics.hpp:491-497); a test item that is also a train item is not a positive.  `noise_zone`: with
    break_ties_with_noise only scores of at least that magnitude count (below it the mt19937 noise separates them, and the
    device reproduces that noise bit for bit)."""
    trp, tri = train[0], train[1]
    tep, tei = test[0], test[1]
    users = np.arange(scores.shape[0]) if users is None else np.asarray(users)
    out = np.zeros(users.shape[0], np.int64)
    n = scores.shape[1]
    for i, u in enumerate(users):
        cand = np.ones(n, bool)
        cand[tri[trp[u]:trp[u + 1]]] = False
        pos = np.unique(tei[tep[u]:tep[u + 1]])
        pos = pos[cand[pos]]
        if pos.size == 0:
            continue
        s = scores[i]
        sp = s[pos]
        if noise_zone is not None:
            sp = sp[np.abs(sp) >= noise_zone]
            if sp.size == 0:
                continue
        sp_sorted = np.sort(sp)
        sc = s[cand]
        lo = np.searchsorted(sp_sorted, sc, "left")
        hi = np.searchsorted(sp_sorted, sc, "right")
        equal = int((hi - lo).sum())              # (candidate, positive) pairs with equal scores, each positive meeting itself once
        out[i] = equal - sp.size
    return out


# ---- the reference's DEFAULT build (-march=native: vectorised, reassociated dot products) --------------------------------------
# SURVEY.md section 8(c), contract item (4): against that build scores differ in their last bits, so a candidate whose score lies
# within rounding of a positive's may change places with it -- and the metrics of that user move by far more than 1e-5 (P@K by
# 1/K).  Every user beyond 1e-5 must be explained by such a near tie; anything else is a real disagreement.
def near_tie_flags(scores, train, test, users, K, eps, k_factors, amax=1.0, rel=2.0 ** -20):
    """For every user of `users` (scores[i] = its exact dense score row): (near_top, near_any) --
    near_any: some positive has a neighbour in the ranking of the candidates whose score is within
              rel * max(|s|, |s'|) + k * eps * amax  (`amax` ~ max|a| * max|b|: what reassociating k products can move);
    near_top: ... and that positive or that neighbour sits among the first K + 1 candidates (what the top-K metrics see)."""
    trp, tri = train[0], train[1]
    tep, tei = test[0], test[1]
    users = np.asarray(users)
    near_top = np.zeros(users.shape[0], bool)
    near_any = np.zeros(users.shape[0], bool)
    n = scores.shape[1]
    floor = float(k_factors) * float(eps) * float(amax)
    for i, u in enumerate(users):
        cand = np.ones(n, bool)
        cand[tri[trp[u]:trp[u + 1]]] = False
        items = np.flatnonzero(cand)
        s = scores[i][items].astype(np.float64)
        s = np.where(np.isnan(s), -np.inf, s)
        order = np.lexsort((items, -s))                    # score desc, item asc (the device's order)
        ss = s[order]
        is_pos = np.zeros(n, bool)
        is_pos[tei[tep[u]:tep[u + 1]]] = True
        pos_sorted = is_pos[items][order]
        gap = np.full(ss.shape[0] + 1, np.inf)
        if ss.shape[0] > 1:
            with np.errstate(invalid="ignore"):
                gap[1:-1] = ss[:-1] - ss[1:]               # gap[r] = between ranks r - 1 and r
                tol = rel * np.maximum(np.abs(ss[:-1]), np.abs(ss[1:])) + floor
            close = np.zeros(ss.shape[0] + 1, bool)
            close[1:-1] = gap[1:-1] <= tol
            # a close pair (r - 1, r) matters when one of the two is a positive and the other is not the same kind of thing
            pair = close[1:-1] & (pos_sorted[:-1] | pos_sorted[1:])
            near_any[i] = bool(pair.any())
            near_top[i] = bool(pair[:min(K + 1, pair.shape[0])].any())
    return near_top, near_any


TOPK_NAMES = ("P@K", "TP@K", "R@K", "AP@K", "TAP@K", "NDCG@K", "Hit@K", "RR@K")


def compare_with_default_build(got, want, scores_of, train, test, K, dtype, k_factors, amax, tol=1e-5, max_explain=256):
    """`got` / `want`: metric name -> array ([users] or [users, K]) from the device and from the reference's vectorised build.
    `scores_of(user_indices)` -> exact dense score rows (rm_debug_scores_*).  Returns the record the bench line and the tests
    print: {users, beyond_1e-5, near_tie_users, unexplained, max_abs_diff, max_abs_diff_unexplained, nan_mismatch}."""
    m = next(iter(got.values())).shape[0]
    beyond_top = np.zeros(m, bool)
    beyond_auc = np.zeros(m, bool)
    nan_mismatch = 0
    worst = 0.0
    per_user_worst = np.zeros(m)
    for name, g in got.items():
        w = want[name]
        g2, w2 = g.reshape(m, -1).astype(np.float64), w.reshape(m, -1).astype(np.float64)
        nan_bad = (np.isnan(g2) != np.isnan(w2)).any(axis=1)
        nan_mismatch += int(nan_bad.sum())
        d = np.abs(np.where(np.isnan(g2) | np.isnan(w2), 0.0, g2 - w2)) / np.maximum(1.0, np.abs(np.where(np.isnan(w2), 0.0, w2)))
        du = d.max(axis=1)
        per_user_worst = np.maximum(per_user_worst, du)
        worst = max(worst, float(du.max(initial=0.0)))
        if name in TOPK_NAMES:
            beyond_top |= (du > tol) | nan_bad
        else:
            beyond_auc |= (du > tol) | nan_bad
    beyond = beyond_top | beyond_auc
    who = np.flatnonzero(beyond)
    rec = {"users": int(m), "beyond_1e-5": int(who.shape[0]), "near_tie_users": 0, "unexplained": 0, "nan_mismatch": int(nan_mismatch),
           "max_abs_diff": worst, "max_abs_diff_unexplained": float(per_user_worst[~beyond].max(initial=0.0)),
           "rule": "a user beyond 1e-5 must have a positive within 2^-20 relative (+ k eps max|a||b|) of a neighbouring candidate's "
                   "score -- among the first K + 1 candidates when a top-K metric differs"}
    if who.shape[0]:
        eps = 2.0 ** -24 if dtype == np.float32 else 2.0 ** -53
        check = who[:max_explain]
        near_top, near_any = near_tie_flags(scores_of(check), train, test, check, K, eps, k_factors, amax)
        explained = np.where(beyond_top[check], near_top, near_any)
        rec["near_tie_users"] = int(explained.sum())
        rec["unexplained"] = int((~explained).sum())
        rec["explained_checked"] = int(check.shape[0])
        if (~explained).any():
            rec["unexplained_users"] = check[~explained][:8].tolist()
            rec["max_abs_diff_unexplained"] = float(per_user_worst[check[~explained]].max())
    rec["ok"] = rec["unexplained"] == 0
    return rec
