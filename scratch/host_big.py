"""Host-pointer calls where the inputs are big (north-star shape, C3's per-GPU shard, C4 / C5 slices): steady time of rm_calc_metrics_*,
with the staged uploads and (RM_DEBUG_NO_STAGED_UPLOAD=1, a child process) without.   python3 scratch/host_big.py [NS C3 C4 C5]"""
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"NS": 32768, "C3": 125000, "C4": 8192, "C5": 16384}


def one(wl):
    from recometrics_amd import _binding as binding
    from recometrics_amd.synth import CONFIGS
    from bench import host_problem
    m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
    m = SHAPES[wl]
    binding.load()
    t0 = time.perf_counter()
    host = host_problem(m, n, k, mean_c, seed, dtype)
    gen = time.perf_counter() - t0
    trp, tri = host["train"]; tep, tei, tev = host["test"]
    want = {name: True for name in binding.METRIC_ORDER}
    cum = wl == "C3"

    def call():
        t = time.perf_counter()
        binding.calc_metrics(host["A"], k, host["B"], k, trp, tri, tep, tei, tev, K, want, cum, False, True, 2, 1, 1, 1)
        return (time.perf_counter() - t) * 1e3
    first = call()
    ts = sorted(call() for _ in range(4))
    nbytes = host["A"].nbytes + host["B"].nbytes + tri.nbytes + tei.nbytes + tev.nbytes + trp.nbytes + tep.nbytes
    print(json.dumps({"workload": wl, "users": m, "staged": not os.environ.get("RM_DEBUG_NO_STAGED_UPLOAD"), "first_ms": first, "steady_ms": ts[len(ts) // 2],
                      "all_ms": ts, "bytes_in": nbytes, "at_45GBs_ms": nbytes / 45e9 * 1e3, "gen_s": gen, "device_ms": binding.timings().get("device_ms")}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for wl in (sys.argv[1:] or list(SHAPES)):
            for env in ({}, {"RM_DEBUG_NO_STAGED_UPLOAD": "1"}):
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", wl], env=e, capture_output=True, text=True, timeout=900)
                print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED %s: %s" % (wl, r.stderr[-300:]))
