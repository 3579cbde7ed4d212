"""Where the first host-pointer call of a process spends its time (BASELINE C2; torch-free): library load, runtime start, a one-user
call (code objects, streams, events), the first full-size call (workspace, page-locked staging, second context), the second.
   python3 scratch/first_call.py [workload]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import host_problem
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
host = host_problem(m, n, k, mean_c, seed, dtype)
trp, tri = host["train"]; tep, tei, tev = host["test"]
want = {name: True for name in binding.METRIC_ORDER}
out = {}
t = time.perf_counter(); binding.load(); out["dlopen_ms"] = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); nd = binding.device_count(); out["device_count_ms"] = (time.perf_counter() - t) * 1e3


def call(users):
    t = time.perf_counter()
    binding.calc_metrics(host["A"][:users], k, host["B"], k, trp[:users + 1], tri, tep[:users + 1], tei, tev, K, want, False, False, True, 2, 1, 1, 1)
    return (time.perf_counter() - t) * 1e3
if os.environ.get("FIRST_TINY"):
    out["tiny_call_ms"] = call(64)
out["first_full_ms"] = call(m)
out["second_full_ms"] = call(m)
out["third_full_ms"] = call(m)
out["steady_ms"] = sorted(call(m) for _ in range(5))[2]
print(json.dumps(out))
