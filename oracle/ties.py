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
