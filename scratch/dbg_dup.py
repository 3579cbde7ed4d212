import numpy as np, sys
from recometrics_amd import _binding as hip
from oracle.oracle import Oracle
from recometrics_amd.synth import make_problem
o = Oracle()
dtype = np.float32
pr = make_problem(200, 3000, 8, dtype, mean_c=60, seed=9)
tep, tei, tev = pr["test"]
rows_i, newp = [], [0]
for u in range(200):
    it = tei[tep[u]:tep[u + 1]]
    rep = np.sort(np.concatenate([it, it, it[:3]]))
    rows_i.append(rep); newp.append(newp[-1] + rep.shape[0])
test = (np.array(newp, np.int32), np.concatenate(rows_i).astype(np.int32), np.ones(newp[-1], dtype))
trp, tri = pr["train"]
for K in (12,):
    want = o.rank(pr["A"], pr["B"], pr["train"], test, K, dtype=dtype, nthreads=8)
    got = hip.rank(pr["A"], pr["B"], trp, tri, test[0], test[1], K)
    bad = np.flatnonzero(got["status"] != want["status"])
    print(K, "status got", np.bincount(got["status"]), "want", np.bincount(want["status"]), bad[:10], [(int(trp[u+1]-trp[u]), int(newp[u+1]-newp[u])) for u in bad[:10]])
    print((got["topk_idx"] == want["topk_idx"]).all(), (got["topk_idx"] == want["topk_idx"]).all(axis=1).sum())
    d = np.flatnonzero(~(got["topk_idx"] == want["topk_idx"]).all(axis=1)); print(d[:5]); [print(got["topk_idx"][x], want["topk_idx"][x], got["topk_score"][x], want["topk_score"][x]) for x in d[:2]]
