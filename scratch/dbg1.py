import sys, numpy as np
sys.path.insert(0, '.')
from oracle.oracle import Oracle
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
o = Oracle()
for (m, n, k, K, c) in [(64, 500, 16, 5, 20), (300, 2000, 32, 7, 40)]:
    pr = make_problem(m, n, k, np.float32, mean_c=c, seed=3)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    w = o.rank(pr["A"], pr["B"], pr["train"], pr["test"], K)
    g = hip.rank(pr["A"], pr["B"], trp, tri, tep, tei, K)
    print("== problem", m, n, k, K, "timings", hip.timings())
    print("status equal:", (w["status"] == g["status"]).all(), "n ranked", (w["status"] == 0).sum())
    bad = np.argwhere((w["topk_idx"] != g["topk_idx"]).any(1)).ravel()
    print("users with topk idx mismatch:", len(bad), bad[:10])
    for u in bad[:3]:
        print(" u", u, "want", w["topk_idx"][u], w["topk_score"][u]); print("     got ", g["topk_idx"][u], g["topk_score"][u])
        print("     train", tri[trp[u]:trp[u+1]][:20], "test", tei[tep[u]:tep[u+1]][:20])
    badr = np.argwhere(w["pos_rank"] != g["pos_rank"]).ravel()
    print("pos_rank mismatches:", len(badr), "of", len(w["pos_rank"]))
    for e in badr[:8]:
        u = np.searchsorted(tep, e, side="right") - 1
        print("  e", e, "user", u, "item", tei[e], "want", w["pos_rank"][e], "got", g["pos_rank"][e], "npos", tep[u+1]-tep[u], "ntr", trp[u+1]-trp[u])
