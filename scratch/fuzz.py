"""Randomised HIP-vs-oracle comparison over shapes and options (scratch tool; the pinned cases live in tests/).
usage (GPU box, repo root): python3 scratch/fuzz.py <cases> <seed>"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
from oracle.oracle import Oracle
from test_hip_parity import hip_calc
from _util import same_bits

hip.load(); oracle = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0; t0 = time.time()
for it in range(cases):
    dtype = np.float32 if rng.random() < 0.65 else np.float64
    m = int(rng.choice([1, 7, 33, 100, 300, 700, 1500]))
    n = int(rng.choice([20, 97, 300, 1111, 4000, 12000]))
    k = int(rng.choice([1, 5, 16, 33, 64, 100, 128, 200, 300]))
    K = int(min(n - 1, rng.choice([1, 3, 10, 20, 33, 60, 100, 256, 300, 600])))
    k = int(rng.choice([k, 500, 520, 1100])) if rng.random() < 0.1 else k
    mean_c = float(min(n / 6, rng.choice([4, 20, 60, 150, 500, 900])))
    allm = ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")
    metrics = allm if rng.random() < 0.6 else tuple(x for x in allm if rng.random() < 0.4) or ("ndcg",)
    if "pr" in metrics and "roc" not in metrics: metrics = metrics + ("roc",)       # deviation D2
    if set(metrics) <= {"hit", "rr"}: metrics = metrics + ("p",)                      # deviation D1
    kw = dict(cumulative=bool(rng.random() < 0.3), cold=bool(rng.random() < 0.7), noise=bool(rng.random() < 0.45), metrics=metrics,
              min_items_pool=int(rng.choice([1, 2, 10])), min_pos_test=int(rng.choice([1, 1, 3])), seed=int(rng.choice([1, 7, 2 ** 35 + 3])))
    pr = make_problem(m, n, k, dtype, mean_c=mean_c, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.3:                           # test items that are also train items (never candidates: reference :491-497)
        trp, tri = pr["train"]; tep_, tei_ = pr["test"][:2]
        rows = []
        for u in range(m):
            tr = tri[trp[u]:trp[u + 1]]; te = tei_[tep_[u]:tep_[u + 1]]
            r = rng.random()
            add = te[rng.random(te.shape[0]) < 0.3] if r < 0.3 else (te if r < 0.36 else te[:0])
            rows.append(np.union1d(tr, add).astype(np.int32))
        pr["train"] = (np.concatenate([[0], np.cumsum([x.shape[0] for x in rows])]).astype(np.int32), np.concatenate(rows).astype(np.int32))
    if kw["noise"] and rng.random() < 0.5:          # cold items: blocks of exactly tied scores that only the noise orders
        pr["B"] = pr["B"].copy(); pr["B"][rng.random(n) < 0.2] = 0
    if os.environ.get("FUZZ_TIES") and not kw["noise"]:   # exact ties without noise (deviation D4: item order): cold items and duplicated items
        pr["B"] = pr["B"].copy(); pr["B"][rng.random(n) < 0.15] = 0
        dup = rng.random(n) < 0.2
        pr["B"][dup] = pr["B"][rng.integers(0, n, int(dup.sum()))]
    if os.environ.get("FUZZ_VERBOSE"):
        print("CASE", it, dict(dtype=dtype.__name__, m=m, n=n, k=k, K=K, mean_c=mean_c, **kw), flush=True)
    try:
        want = oracle.calc(pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, nthreads=8, **kw)
        got = hip_calc(hip, pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, **kw)
        tep = pr["test"][0]
        one_chunk = np.diff(tep) <= 63
        msg = []
        for name in want:
            g, w = got[name], want[name]
            nan_ok = (np.isnan(g) == np.isnan(w)).all()
            w0 = np.where(np.isnan(w), 0, w).astype(np.float64)
            d = np.nanmax(np.abs(np.where(np.isnan(g), 0, g).astype(np.float64) - w0) / np.maximum(1.0, np.abs(w0))) if g.size else 0.0
            if not nan_ok or d > 1e-5:
                msg.append("%s nan_ok=%s maxdiff=%g" % (name, nan_ok, d))
            elif name == "PR_AUC" and "RM_STREAM_BUDGET_MB" in os.environ:      # chunked sums of long rows: 1e-12
                if d > 1e-12: msg.append("PR_AUC maxdiff=%g" % d)
            elif name != "ROC_AUC" and not same_bits(g, w).all():
                msg.append("%s not bitwise (%d)" % (name, (~same_bits(g, w)).sum()))
        rkw = dict(noise=kw["noise"], seed=kw["seed"], cold=kw["cold"], min_items_pool=kw["min_items_pool"], min_pos_test=kw["min_pos_test"])
        wr = oracle.rank(pr["A"], pr["B"], pr["train"], pr["test"], K, dtype=dtype, nthreads=8, **rkw)
        trp, tri = pr["train"]; tei = pr["test"][1]
        gr = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, K,
                      break_ties_with_noise=kw["noise"], seed=kw["seed"], consider_cold_start=kw["cold"], min_items_pool=kw["min_items_pool"],
                      min_pos_test=kw["min_pos_test"])
        if not (gr["status"] == wr["status"]).all(): msg.append("status")
        if not (gr["topk_idx"] == wr["topk_idx"]).all(): msg.append("topk_idx")
        if not (gr["pos_rank"] == wr["pos_rank"]).all(): msg.append("pos_rank")
    except Exception as e:      # noqa: BLE001
        msg = ["exception %r" % (e,)]
    if msg:
        bad += 1
        print("FAIL", dict(dtype=dtype.__name__, m=m, n=n, k=k, K=K, mean_c=mean_c, **kw), msg, flush=True)
print("fuzz: %d cases, %d failed, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
