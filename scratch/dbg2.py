import sys, numpy as np
sys.path.insert(0, '.')
from oracle.oracle import Oracle
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
o = Oracle()
for (m, n, k, K, c, seed) in [(64, 40000, 128, 10, 100, 64+40000), (500, 3000, 16, 40, 60, 3500), (96, 6000, 24, 10, 700, 77)]:
    pr = make_problem(m, n, k, np.float32, mean_c=c, seed=seed)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    w = o.rank(pr["A"], pr["B"], pr["train"], pr["test"], K, nthreads=8)
    g = hip.rank(pr["A"], pr["B"], trp, tri, tep, tei, K)
    print("== problem", m, n, k, K, "timings", hip.timings())
    print("status equal:", (w["status"] == g["status"]).all(), "n ranked", (w["status"] == 0).sum())
    bad = np.argwhere((w["topk_idx"] != g["topk_idx"]).any(1)).ravel()
    print("users with topk idx mismatch:", len(bad), bad[:10])
    badr = np.argwhere(w["pos_rank"] != g["pos_rank"]).ravel()
    print("pos_rank mismatches:", len(badr), "of", len(w["pos_rank"]))
    users = np.searchsorted(tep, badr, side="right") - 1
    print(" bad users:", np.unique(users)[:20], " their npos:", [int(tep[u+1]-tep[u]) for u in np.unique(users)[:20]])
    npos = np.diff(tep); print(" npos max", npos.max(), " users with npos>63:", (npos > 63).sum())
    for e in badr[:6]:
        u = np.searchsorted(tep, e, side="right") - 1
        print("  e", e, "user", u, "item", tei[e], "want", w["pos_rank"][e], "got", g["pos_rank"][e], "npos", tep[u+1]-tep[u], "ntr", trp[u+1]-trp[u])
