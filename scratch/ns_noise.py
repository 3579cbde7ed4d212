"""C2 (or another workload) with break_ties_with_noise=True: python3 scratch/ns_noise.py [workload] [users] [steps]"""
import sys, os, json, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
m = int(sys.argv[2]) if len(sys.argv) > 2 else m
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype)
st = torch.cuda.current_stream().cuda_stream
p.step(binding, st, noise=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): p.step(binding, st, noise=True)
torch.cuda.synchronize()
print(json.dumps({"workload": wl, "users": m, "ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "tm_last_pass": binding.timings()}))
