import sys, os, json, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem, measure
import torch.distributed as dist
wl = sys.argv[1] if len(sys.argv) > 1 else "NS"
users = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
m = users
K = int(os.environ.get('NS_K', K))
k = int(os.environ.get('NS_FACTORS', k))
n = int(os.environ.get('NS_ITEMS', n))
if os.environ.get('NS_DTYPE'): dtype = {'f32': np.float32, 'f64': np.float64}[os.environ['NS_DTYPE']]
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
cumulative = wl == "C3" or bool(os.environ.get("NS_CUMULATIVE"))          # C3 is quoted with K = 1..20 (cumulative)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype, cumulative=cumulative)
p.noise = bool(os.environ.get("NS_NOISE"))
if os.environ.get("NS_TEST_DIV"):       # experiment: the test items squeezed into the first n/DIV items (rows stay sorted; repeats allowed)
    p.tei = torch.div(p.tei, int(os.environ["NS_TEST_DIV"]), rounding_mode="floor").to(p.tei.dtype)
if os.environ.get("NS_DROP"):            # experiment: metrics that are not asked for (indices into METRIC_ORDER, comma separated)
    drop = [int(x) for x in os.environ["NS_DROP"].split(",")]
    full = p.out_ptrs
    p.out_ptrs = lambda o: [0 if i in drop else q for i, q in enumerate(full(o))]
dt, sw, pr, fi, tm = measure(torch, dist, binding, p, steps, 1, 1, None)
tf = 2.0 * n * k * m / (sw * 1e-3) / 1e12
peak = 157.3 if dtype == np.float32 else 78.6
print(json.dumps({"workload": wl, "users": m, "users_per_s": m * steps / dt, "sweep_ms": sw, "prep_ms": pr, "fin_ms": fi, "TF": tf, "frac": tf / peak, "tm": tm}))
if os.environ.get("RM_PRINT_STATS"):
    import ctypes
    lib = binding.load()
    lib = ctypes.CDLL(os.environ["RECOMETRICS_HIP_LIB"])
    buf = (ctypes.c_ulonglong * 16)()
    getattr(lib, os.environ.get("RM_STATS_FN", "rm_debug_stats"))(buf, 1)
    runs = steps + 1
    names = ["full_passes", "sample_passes", "tie_events", "below_events", "append_cycles", "merges_or_selections", "merge_iters_or_selection_cycles", "barrier_wait_cycles", "cyc_loop_g3s0", "cyc_wait_sync", "cyc_prologue", "cyc_loop", "cyc_epilogue", "tiles", "cyc_loop_sub1", "cyc_loop_sub2"]
    print(json.dumps({nm: buf[i] / runs for i, nm in enumerate(names)}))
