# PCIe-inclusive rate: the host-pointer C-ABI entry (stages A, B, CSR to HBM, runs, copies the metrics back)
import sys, os, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import _binding as b
from recometrics_amd.synth import CONFIGS, make_factors, make_interactions
m, n, k, dtype, K, c, seed = CONFIGS["C2"]
A, B = make_factors(m, n, k, dtype, seed)
trp, tri, tep, tei, tev = make_interactions(m, n, c, dtype, seed)
want = {nm: True for nm in b.METRIC_ORDER}
for rep in range(3):
    t0 = time.perf_counter()
    out = b.calc_metrics(A, k, B, k, trp, tri, tep, tei, tev, K, want, False, False, True, 2, 1, 1, 1)
    dt = time.perf_counter() - t0
    print(json.dumps({"rep": rep, "host_api_s": dt, "users_per_s": m / dt, "bytes_in": A.nbytes + B.nbytes + trp.nbytes*2 + tri.nbytes + tei.nbytes + tev.nbytes, "timings": b.timings()}))
