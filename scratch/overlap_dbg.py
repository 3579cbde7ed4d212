import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
from oracle.oracle import Oracle
from test_hip_parity import hip_calc
hip.load(); oracle = Oracle()
dtype = np.float64
rng = np.random.default_rng(4242)
pr = make_problem(260, 3000 + 11, 40, dtype, mean_c=220, seed=88)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
rows = []
for u in range(trp.shape[0] - 1):
    tr = tri[trp[u]:trp[u + 1]]; te = tei[tep[u]:tep[u + 1]]
    if u % 3 == 0: add = te[rng.random(te.shape[0]) < 0.3]
    elif u % 11 == 1: add = te
    else: add = te[:0]
    rows.append(np.union1d(tr, add).astype(np.int32))
trp2 = np.concatenate([[0], np.cumsum([r.shape[0] for r in rows])]).astype(np.int32)
pr["train"] = (trp2, np.concatenate(rows).astype(np.int32))
for env in ({}, {"RM_STREAM_BUDGET_MB": "0"}, {"RM_DEBUG_NO_SIDE": "1"}):
    for kk in ("RM_STREAM_BUDGET_MB", "RM_DEBUG_NO_SIDE"): os.environ.pop(kk, None)
    os.environ.update(env)
    w = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype, nthreads=8, noise=False, seed=5)
    g = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], 10, dtype=dtype, noise=False, seed=5)
    print(env)
    for name in w:
        a = np.nan_to_num(w[name].astype(np.float64)); b = np.nan_to_num(g[name].astype(np.float64))
        d = np.abs(a - b).reshape(260, -1).max(axis=1)
        bad = np.argwhere(d > 1e-9).ravel()
        if bad.size:
            print(" ", name, "maxdiff", d.max(), "nan mismatch", (np.isnan(w[name]) != np.isnan(g[name])).sum(), "users", bad[:12])
            for u in bad[:6]:
                te = tei[tep[u]:tep[u + 1]]
                print("    user", u, "ntest", te.shape[0], "ntrain", rows[u].shape[0], "overlap", np.intersect1d(rows[u], te).shape[0], "want", w[name][u], "got", g[name][u])
