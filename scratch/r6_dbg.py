import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
from oracle.oracle import Oracle
hip.load(); oracle = Oracle()
m, n, k, K, mean_c = [int(x) for x in sys.argv[1:6]]
dtype = np.float32
for seed in range(5):
    pr = make_problem(m, n, k, dtype, mean_c=float(mean_c), seed=1234 + seed)
    trp, tri = pr["train"]; tep, tei = pr["test"][:2]
    wr = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, nthreads=8, cold=False, min_items_pool=10)
    gr = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, K, consider_cold_start=False, min_items_pool=10)
    bad = np.flatnonzero((gr["topk_idx"] != wr["topk_idx"]).any(axis=1))
    print("seed", seed, "bad users", bad.shape[0], "of", m)
    for u in bad[:4]:
        npos = tep[u + 1] - tep[u]; ntr = trp[u + 1] - trp[u]
        print(" user", u, "npos", npos, "ntr", ntr, "want", wr["topk_idx"][u], wr["topk_score"][u], "got", gr["topk_idx"][u], gr["topk_score"][u])
        te = tei[tep[u]:tep[u + 1]]
        print("   want in test:", np.isin(wr["topk_idx"][u], te), "got in test:", np.isin(gr["topk_idx"][u], te))
    if bad.shape[0]: break
