"""Synthetic MovieLens-/MSD-shaped inputs for tests and bench.py.

Factors ``A, B ~ N(0,1)/sqrt(k)`` (scores ~ N(0,1), tie-free with overwhelming probability);
per-user interaction counts ``clip(round(lognormal(mu, 1)), 2, n/4)`` with the requested mean;
items uniform without replacement, sorted; ``max(1, round(0.3*c))`` of them held out as the test
row (the reference's default ``items_test_fraction``, recometrics/__init__.py:635), the rest is
the train row; test values uniform in {1..20}.
"""
import numpy as np

CONFIGS = {
    # name: (m, n, k, dtype, K, mean interactions, seed)
    "C1": (1_000, 5_000, 64, np.float32, 10, 50, 101),
    "C2": (138_493, 26_744, 64, np.float32, 10, 144, 102),
    "C3": (1_000_000, 380_000, 128, np.float32, 20, 48, 103),
    "C4": (100_000, 10_000_000, 128, np.float32, 100, 10, 104),
    "C5": (200_000, 500_000, 256, np.float64, 50, 50, 105),
    "NS": (32_768, 1_000_000, 128, np.float32, 10, 100, 100),
    # the only realistic workload the reference documents (examples/recometrics_example.ipynb cells 3, 5, 9, 11; BASELINE.md
    # section 1): LastFM-360K -- 10,000 test users x 160,112 items, 50 factors, k = 5, all metrics, API defaults (noise on)
    "TUT": (10_000, 160_112, 50, np.float32, 5, 50, 108),
    # probes (not BASELINE configs): fp64 with resident user factors / small K
    "P64a": (8_192, 500_000, 128, np.float64, 10, 50, 106),
    "P64b": (8_192, 500_000, 256, np.float64, 10, 50, 107),
}


def make_factors(m, n, k, dtype=np.float32, seed=0):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((m, k), dtype=np.float32) / np.sqrt(k)).astype(dtype)
    B = (rng.standard_normal((n, k), dtype=np.float32) / np.sqrt(k)).astype(dtype)
    return A, B


def make_interactions(m, n, mean_c, dtype=np.float32, seed=0, test_fraction=0.3):
    """Returns (train_indptr, train_indices, test_indptr, test_indices, test_values)."""
    rng = np.random.default_rng(seed + 7919)
    mu = np.log(max(mean_c, 1.0)) - 0.5
    c = np.clip(np.rint(rng.lognormal(mu, 1.0, size=m)), 2, max(2, n // 4)).astype(np.int64)
    # draw with a little slack, dedupe per user
    tr_p = np.zeros(m + 1, dtype=np.int64)
    te_p = np.zeros(m + 1, dtype=np.int64)
    tr_chunks, te_chunks = [], []
    for u in range(m):
        cu = int(c[u])
        items = np.unique(rng.integers(0, n, size=cu + (cu >> 3) + 2))
        if items.shape[0] > cu:
            items = np.sort(rng.permutation(items)[:cu])
        cu = items.shape[0]
        nte = min(max(1, int(round(test_fraction * cu))), cu)
        pick = np.zeros(cu, dtype=bool)
        pick[rng.permutation(cu)[:nte]] = True
        te_chunks.append(items[pick])
        tr_chunks.append(items[~pick])
        te_p[u + 1] = te_p[u] + nte
        tr_p[u + 1] = tr_p[u] + (cu - nte)
    tr_i = np.concatenate(tr_chunks).astype(np.int32) if tr_chunks else np.zeros(0, np.int32)
    te_i = np.concatenate(te_chunks).astype(np.int32) if te_chunks else np.zeros(0, np.int32)
    te_v = rng.integers(1, 21, size=te_i.shape[0]).astype(dtype)
    return tr_p.astype(np.int32), tr_i, te_p.astype(np.int32), te_i, te_v


def make_interactions_fast(m, n, mean_c, dtype=np.float32, seed=0, test_fraction=0.3):
    """The same kind of data as make_interactions (lognormal row lengths, uniform items, ~30 % of a row held out, at least one test
    item per user), drawn in whole-array operations: seconds instead of minutes for a million users.  NOT the same draws -- the
    workloads bench.py times keep make_interactions; this one feeds the tests that run a BASELINE config at its full user count."""
    rng = np.random.default_rng(seed + 104729)
    mu = np.log(max(mean_c, 1.0)) - 0.5
    c = np.clip(np.rint(rng.lognormal(mu, 1.0, size=m)), 2, max(2, n // 4)).astype(np.int64)
    users = np.repeat(np.arange(m, dtype=np.int64), c)
    key = np.unique(users * np.int64(n) + rng.integers(0, n, size=users.shape[0], dtype=np.int64))      # sorted by (user, item), duplicates dropped
    users, items = key // n, (key % n).astype(np.int32)
    del key
    start = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(users, minlength=m), out=start[1:])
    held = rng.random(users.shape[0]) < test_fraction
    none = np.bincount(users[held], minlength=m) == 0                    # users without a test item: their first entry becomes one
    held[start[:-1][none & (start[1:] > start[:-1])]] = True
    all_held = (np.bincount(users[~held], minlength=m) == 0) & (start[1:] - start[:-1] > 1)      # ... and nobody loses the whole row
    held[start[:-1][all_held]] = False
    te_p = np.zeros(m + 1, np.int64); tr_p = np.zeros(m + 1, np.int64)
    np.cumsum(np.bincount(users[held], minlength=m), out=te_p[1:])
    np.cumsum(np.bincount(users[~held], minlength=m), out=tr_p[1:])
    te_i, tr_i = items[held], items[~held]
    te_v = rng.integers(1, 21, size=te_i.shape[0]).astype(dtype)
    return tr_p.astype(np.int32), tr_i, te_p.astype(np.int32), te_i, te_v


def make_problem(m, n, k, dtype=np.float32, mean_c=50, seed=0):
    A, B = make_factors(m, n, k, dtype, seed)
    trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed)
    return {"A": A, "B": B, "train": (trp, tri), "test": (tep, tei, tev)}
