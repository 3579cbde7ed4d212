import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
from recometrics_amd import _binding as b
from recometrics_amd.synth import CONFIGS, make_factors, make_interactions
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
_, B = make_factors(1, n, k, dtype, seed); A, _ = make_factors(m, 1, k, dtype, seed + 1)
trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed)
want = {name: True for name in b.METRIC_ORDER}
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    b.calc_metrics(A, k, B, k, trp, tri, tep, tei, tev, K, want, False, False, True, 2, 1, 1, 1)
    ts.append((time.perf_counter() - t0) * 1e3)
print(sys.argv[1:], "min %.3f median %.3f ms" % (min(ts[1:]), sorted(ts[1:])[2]), flush=True)
os.system("grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u | head -5" % os.getpid())
