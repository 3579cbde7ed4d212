"""Reproduces the first failing case of scratch/fuzz.py <cases> <seed> and prints the mismatching positives."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
from oracle.oracle import Oracle
hip.load(); oracle = Oracle()
rng = np.random.default_rng(int(sys.argv[2]))
for it in range(int(sys.argv[1])):
    dtype = np.float32 if rng.random() < 0.65 else np.float64
    m = int(rng.choice([1, 7, 33, 100, 300, 700, 1500])); n = int(rng.choice([20, 97, 300, 1111, 4000, 12000]))
    k = int(rng.choice([1, 5, 16, 33, 64, 100, 128, 200, 300])); K = int(min(n - 1, rng.choice([1, 3, 10, 20, 33, 60, 100, 256, 300, 600])))
    k = int(rng.choice([k, 500, 520, 1100])) if rng.random() < 0.1 else k
    mean_c = float(min(n / 6, rng.choice([4, 20, 60, 150, 500, 900])))
    allm = ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")
    metrics = allm if rng.random() < 0.6 else tuple(x for x in allm if rng.random() < 0.4) or ("ndcg",)
    kw = dict(cumulative=bool(rng.random() < 0.3), cold=bool(rng.random() < 0.7), noise=bool(rng.random() < 0.45), metrics=metrics,
              min_items_pool=int(rng.choice([1, 2, 10])), min_pos_test=int(rng.choice([1, 1, 3])), seed=int(rng.choice([1, 7, 2 ** 35 + 3])))
    pr = make_problem(m, n, k, dtype, mean_c=mean_c, seed=int(rng.integers(1 << 30)))
    if kw["noise"] and rng.random() < 0.5:
        pr["B"] = pr["B"].copy(); pr["B"][rng.random(n) < 0.2] = 0
    if not (kw["noise"] and dtype == np.float32 and n == 12000):
        continue
    rkw = dict(noise=kw["noise"], seed=kw["seed"], cold=kw["cold"], min_items_pool=kw["min_items_pool"], min_pos_test=kw["min_pos_test"])
    wr = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, nthreads=8, **rkw)
    trp, tri = pr["train"]; tep, tei = pr["test"][:2]
    gr = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, K, break_ties_with_noise=True,
                  seed=kw["seed"], consider_cold_start=kw["cold"], min_items_pool=kw["min_items_pool"], min_pos_test=kw["min_pos_test"])
    bad = np.nonzero(gr["pos_rank"] != wr["pos_rank"])[0]
    if bad.size == 0:
        continue
    print("case", it, dict(m=m, n=n, k=k, K=K, mean_c=mean_c, **kw))
    users = np.searchsorted(tep, bad, side="right") - 1
    S = oracle.scores(pr["A"][np.unique(users)], pr["B"])
    for e, u in zip(bad[:6], users[:6]):
        row = S[list(np.unique(users)).index(u)]
        item = tei[e]
        print(" user", u, "entry", e, "item", item, "score", repr(row[item]), "P", tep[u + 1] - tep[u], "ntr", trp[u + 1] - trp[u],
              "rank hip/oracle", gr["pos_rank"][e], wr["pos_rank"][e], "zero-score items", int((row == 0).sum()), "status", gr["status"][u], wr["status"][u])
    break

# ---- emulate the reference's noise for the first mismatching user and list what ties with the positive ----
def init_mt(seed):
    x = np.zeros(624, np.uint32); x[0] = seed & 0xffffffff
    for i in range(1, 624):
        x[i] = (1812433253 * (int(x[i - 1]) ^ (int(x[i - 1]) >> 30)) + i) & 0xffffffff
    return x
u = int(users[0]); e = int(bad[0]); item = int(tei[e])
row = S[list(np.unique(users)).index(u)].astype(np.float32)
mask = np.ones(n, bool); mask[tri[trp[u]:trp[u + 1]]] = False
bg = np.random.MT19937(); st = bg.state; st['state']['key'] = init_mt(kw["seed"] + u); st['state']['pos'] = 624; bg.state = st
d = bg.random_raw(int(mask.sum())).astype(np.uint32)
r = d.astype(np.float32) * np.float32(2.0 ** -32); r[r >= 1] = np.float32(0.99999994)
ee = (r * (np.float32(1e-12) - np.float32(-1e-12))).astype(np.float32) + np.float32(-1e-12)
sn = row.copy(); sn[mask] = (row[mask] + ee).astype(np.float32)
same = np.nonzero(mask & (sn == sn[item]))[0]
pos_items = set(tei[tep[u]:tep[u + 1]].tolist())
print(" noisy score of the positive", repr(sn[item]), "items with exactly that noisy score:", [(int(j), "positive" if int(j) in pos_items else "candidate") for j in same])
above = int((sn[mask] > sn[item]).sum())
print(" candidates strictly above:", above, "-> rank by (score desc, item asc):", above + 1 + int(sum(1 for j in same if j < item)))

# ---- the same user alone, and in small groups ----
from recometrics_amd.sharding import slice_csr
for lo, hi in ((u, u + 1), (max(0, u - 40), u + 40), (0, m)):
    hi = min(hi, m)
    stp, sti, _ = slice_csr(trp, tri, None, lo, hi)
    sep, sei, _ = slice_csr(tep, tei, None, lo, hi)
    for env in ({}, {"RM_STREAM_BUDGET_MB": "0"}, {"RM_DEBUG_NO_TRAIN_BITS": "1"}, {"RM_DEBUG_SPLITS": "1"}, {"RM_DEBUG_SPLITS": "3"}):
        for kk in ("RM_STREAM_BUDGET_MB", "RM_DEBUG_NO_TRAIN_BITS", "RM_DEBUG_SPLITS"): os.environ.pop(kk, None)
        os.environ.update(env)
        g2 = hip.rank(np.ascontiguousarray(pr["A"][lo:hi]), np.ascontiguousarray(pr["B"]), stp, sti, sep, sei, K, break_ties_with_noise=True,
                      seed=kw["seed"] + lo, consider_cold_start=kw["cold"], min_items_pool=kw["min_items_pool"], min_pos_test=kw["min_pos_test"])
        a = e - tep[lo]
        print(" users [%d,%d) env %s: rank of the entry %d (oracle %d)" % (lo, hi, env, g2["pos_rank"][a], wr["pos_rank"][e]))

# ---- the device's own noisy scores of the two tied items: full ranking of the user alone (k_metrics = candidates - 1) ----
for kk in ("RM_STREAM_BUDGET_MB", "RM_DEBUG_NO_TRAIN_BITS", "RM_DEBUG_SPLITS"): os.environ.pop(kk, None)
stp, sti, _ = slice_csr(trp, tri, None, u, u + 1)
sep, sei, _ = slice_csr(tep, tei, None, u, u + 1)
Kf = int(mask.sum()) - 1
g3 = hip.rank(np.ascontiguousarray(pr["A"][u:u + 1]), np.ascontiguousarray(pr["B"]), stp, sti, sep, sei, Kf, break_ties_with_noise=True,
              seed=kw["seed"] + u, consider_cold_start=True, min_items_pool=1, min_pos_test=1)
idx, sc = g3["topk_idx"][0], g3["topk_score"][0]
for j in [int(x) for x in same]:
    at = np.nonzero(idx == j)[0]
    print(" item", j, "device position", at, "device noisy score", [repr(sc[a]) for a in at], "emulated", repr(sn[j]), "bits", hex(int(sn[j:j+1].view(np.uint32)[0])),
          "device bits", [hex(int(sc[a:a+1].view(np.uint32)[0])) for a in at])
diff = [(int(i), repr(s), repr(sn[i])) for i, s in zip(idx, sc) if i >= 0 and s != sn[i]]
print(" items whose device noisy score differs from the emulation:", len(diff), diff[:5])
