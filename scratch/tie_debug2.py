import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from oracle.oracle import Oracle
hip.load(); oracle = Oracle()
rng = np.random.default_rng(5)
n, k, m = 12000, 8, 600
A = rng.standard_normal((m, k)).astype(np.float32); B = rng.standard_normal((n, k)).astype(np.float32)
B[rng.random(n) < 0.25] = 0
trp = np.zeros(m + 1, np.int32); tri = np.zeros(0, np.int32)
rows = [np.unique(rng.permutation(n)[:(300 if u % 2 == 0 else 50)]) for u in range(m)]
tep = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32); tei = np.concatenate(rows).astype(np.int32)
for env in ({}, {"RM_STREAM_BUDGET_MB": "0"}):
    os.environ.pop("RM_STREAM_BUDGET_MB", None); os.environ.update(env)
    wr = oracle.rank(A, B, (trp, tri), (tep, tei), 10, nthreads=8, noise=True, seed=7)
    gr = hip.rank(A, B, trp, tri, tep, tei, 10, break_ties_with_noise=True, seed=7)
    bad = np.nonzero(gr["pos_rank"] != wr["pos_rank"])[0]
    users = np.searchsorted(tep, bad, side="right") - 1
    print(env, "mismatching entries", bad.size, "users (even = 300 positives, odd = 50):", sorted(set(users.tolist()))[:30])
    for e, u in list(zip(bad, users))[:6]:
        print("  user", u, "P", tep[u+1]-tep[u], "item", tei[e], "hip/oracle", gr["pos_rank"][e], wr["pos_rank"][e])
