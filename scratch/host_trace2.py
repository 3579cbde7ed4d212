import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import _binding as b
from recometrics_amd.synth import CONFIGS, make_factors, make_interactions
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
_, B = make_factors(1, n, k, dtype, seed); A, _ = make_factors(m, 1, k, dtype, seed + 1)
trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed)
want = {name: True for name in b.METRIC_ORDER}
envs = [{}, {"RM_DEBUG_RAMP": "6"}, {"RM_DEBUG_RAMP": "12"}, {"RM_DEBUG_RAMP": "16"}]
for env in envs:
    for kk in ("RM_DEBUG_RAMP",): os.environ.pop(kk, None)
    for kk, vv in env.items(): os.environ[kk] = vv
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        b.calc_metrics(A, k, B, k, trp, tri, tep, tei, tev, K, want, False, False, True, 2, 1, 1, 1)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(env, "min %.3f  median %.3f ms" % (min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2]), flush=True)
os.environ.pop("RM_DEBUG_RAMP", None)
os.environ["RM_HOST_TRACE"] = "1"
for rep in range(2):
    b.calc_metrics(A, k, B, k, trp, tri, tep, tei, tev, K, want, False, False, True, 2, 1, 1, 1)
