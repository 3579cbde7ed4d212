"""Where the time of the call a user makes goes (C2, noise on, SciPy CSR in): python3 scratch/api_profile.py"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp
import recometrics_amd
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import host_problem
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
host = host_problem(m, n, k, mean_c, seed, dtype)
trp, tri = host["train"]; tep, tei, tev = host["test"]
ones = np.ones(tri.shape[0], dtype)
fresh = lambda: (sp.csr_array((ones, tri, trp), shape=(m, n), copy=False), sp.csr_array((tev, tei, tep), shape=(m, n), copy=False))
want = {name: True for name in binding.METRIC_ORDER}
def host_call(noise):
    t0 = time.perf_counter()
    binding.calc_metrics(host["A"], k, host["B"], k, trp, tri, tep, tei, tev, K, want, False, noise, True, 2, 1, 1, 1)
    return (time.perf_counter() - t0) * 1e3
def api_call(mats, **kw):
    t0 = time.perf_counter()
    recometrics_amd.calc_reco_metrics(mats[0], mats[1], host["A"], host["B"], k=K, all_metrics=True, as_df=False, **kw)
    return (time.perf_counter() - t0) * 1e3
for noise in (False, True):
    host_call(noise)
    print("host entry noise=%s:" % noise, sorted(round(host_call(noise), 2) for _ in range(7)))
api_call(fresh())
print("api fresh objects, noise on :", sorted(round(api_call(fresh()), 2) for _ in range(7)))
print("api fresh objects, noise off:", sorted(round(api_call(fresh(), break_ties_with_noise=False), 2) for _ in range(7)))
mats = fresh(); api_call(mats)
print("api same objects,  noise on :", sorted(round(api_call(mats), 2) for _ in range(7)))
for nt in (1, 4, 16, 64, 0):
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); binding.csr_rows_sorted(trp, tri, nt); binding.csr_rows_sorted(tep, tei, nt); t.append((time.perf_counter() - t0) * 1e3)
    print("sortedness check of both matrices, %d threads: %.2f ms" % (nt, sorted(t)[2]))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    api_call(fresh())
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
os.environ["RM_HOST_TRACE"] = "1"
binding.reload_switches()
host_call(False); host_call(True)
