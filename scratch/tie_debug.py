import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from oracle.oracle import Oracle
hip.load(); oracle = Oracle()
rng = np.random.default_rng(3)
n, k, m = 3000, 8, 40
A = rng.standard_normal((m, k)).astype(np.float32); B = rng.standard_normal((n, k)).astype(np.float32)
# duplicate item rows: item j and j + 1500 have identical factors for j < 200 -> exact score ties for every user
B[1500:1700] = B[0:200]
trp = np.zeros(m + 1, np.int32); tri = np.zeros(0, np.int32)
rows = []
for u in range(m):
    P = 100 if u % 2 == 0 else 30            # streamed (P > 63) and table users
    items = rng.permutation(n)[:P]
    # make sure some positives are members of tied pairs, on either side
    items[:10] = rng.permutation(200)[:10] + (1500 if u % 4 < 2 else 0)
    rows.append(np.unique(items))
tep = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32); tei = np.concatenate(rows).astype(np.int32)
for noise in (False,):
    wr = oracle.rank(A, B, (trp, tri), (tep, tei), 10, nthreads=4, noise=noise)
    gr = hip.rank(A, B, trp, tri, tep, tei, 10, break_ties_with_noise=noise)
    bad = np.nonzero(gr["pos_rank"] != wr["pos_rank"])[0]
    users = np.searchsorted(tep, bad, side="right") - 1
    print("noise", noise, "mismatching entries", bad.size, "users", sorted(set(users.tolist()))[:20])
    for e, u in list(zip(bad, users))[:8]:
        print("  user", u, "P", tep[u+1]-tep[u], "item", tei[e], "hip/oracle", gr["pos_rank"][e], wr["pos_rank"][e])
