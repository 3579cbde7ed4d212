import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
hip.load()
K = int(sys.argv[1]); k = int(sys.argv[2]); dtype = np.float32 if sys.argv[3] == "f32" else np.float64; noise = sys.argv[4] == "1"
m = int(sys.argv[5]) if len(sys.argv) > 5 else 300; n = int(sys.argv[6]) if len(sys.argv) > 6 else 12000
pr = make_problem(m, n, k, dtype, mean_c=min(150.0, n / 6), seed=600 + K)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
want = {name: True for name in hip.METRIC_ORDER}
got = hip.calc_metrics(pr["A"], k, pr["B"], k, trp, tri, tep, tei, tev, K, want, False, noise, True, 1, 1, 1, 7)
print("ok", sys.argv[1:], float(np.nanmean(got[0])))
