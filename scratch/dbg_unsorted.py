"""debug: unsorted rows through the host entry"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
hip.load()
pr = make_problem(900, 5000, 32, np.float32, mean_c=70, seed=11)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
want = {name: True for name in hip.METRIC_ORDER}
def call(tri, tei, tev):
    return hip.calc_metrics(pr["A"], 32, pr["B"], 32, trp, tri, tep, tei, tev, 10, want, False, False, True, 2, 1, 1, 5)
a = call(tri, tei, tev)
print("sorted ok", flush=True)
tri2 = tri.copy()
u = 5
tri2[trp[u]:trp[u + 1]] = tri2[trp[u]:trp[u + 1]][::-1]
b = call(tri2, tei, tev)
print("unsorted ok", all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), flush=True)
