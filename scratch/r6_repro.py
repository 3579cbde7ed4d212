import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
hip.load()
K = int(sys.argv[1]); k = int(sys.argv[2]); dtype = np.float32 if sys.argv[3] == "f32" else np.float64
pr = make_problem(140, 12000, k, dtype, mean_c=70, seed=600 + K)
trp, tri = pr["train"]; tep, tei = pr["test"][:2]
got = hip.rank(np.ascontiguousarray(pr["A"], dtype), np.ascontiguousarray(pr["B"], dtype), trp, tri, tep, tei, K)
print("ok", K, k, sys.argv[3], got["topk_idx"][0][:5])
