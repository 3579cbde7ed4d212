"""host-pointer entry at C2 under different first-batch fractions (RM_DEBUG_RAMP): python3 scratch/host_ramp.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import host_problem
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
host = host_problem(m, n, k, mean_c, seed, dtype)
trp, tri = host["train"]; tep, tei, tev = host["test"]
want = {name: True for name in binding.METRIC_ORDER}
def host_call(noise):
    t0 = time.perf_counter()
    binding.calc_metrics(host["A"], k, host["B"], k, trp, tri, tep, tei, tev, K, want, False, noise, True, 2, 1, 1, 1)
    return (time.perf_counter() - t0) * 1e3
for ramp in (None, "2", "3", "4", "5", "6", "8", "12"):
    if ramp is None: os.environ.pop("RM_DEBUG_RAMP", None)
    else: os.environ["RM_DEBUG_RAMP"] = ramp
    binding.reload_switches()
    for noise in (False, True):
        host_call(noise); host_call(noise)
        t = sorted(round(host_call(noise), 2) for _ in range(9))
        print("ramp", ramp, "noise", noise, "median %.2f min %.2f" % (t[4], t[0]), t, flush=True)
os.environ.pop("RM_DEBUG_RAMP", None); os.environ["RM_HOST_TRACE"] = "1"; binding.reload_switches()
host_call(False)
