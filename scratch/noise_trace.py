import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
if len(sys.argv) > 2: m = int(sys.argv[2])
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype)
st = torch.cuda.current_stream().cuda_stream
for rnd in range(2):
  for noise, env in ((False, {}), (True, {}), (True, {"RM_DEBUG_NO_EXT_BITS": "1"}), (True, {"RM_DEBUG_NOISE_SEQUENTIAL": "1"})):
    for kk in ("RM_DEBUG_NO_EXT_BITS", "RM_DEBUG_NOISE_SEQUENTIAL"): os.environ.pop(kk, None)
    os.environ.update(env)
    p.step(binding, st, noise=noise); p.step(binding, st, noise=noise); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): p.step(binding, st, noise=noise)
    torch.cuda.synchronize()
    print("noise", noise, env, "%.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
