import sys, os, time, subprocess
# noise-on step at C2 with / without the high-priority peer stream (one process each: the stream is created once per process)
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype)
st = torch.cuda.current_stream().cuda_stream
for noise in (False, True):
    for _ in range(3): p.step(binding, st, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): p.step(binding, st, noise=noise)
    torch.cuda.synchronize()
    print("noise", noise, "%.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
'''
for rnd in range(2):
    for env in ({}, {"RM_DEBUG_NOISE_NO_PRIORITY": "1"}, {"RM_DEBUG_NOISE_SEQUENTIAL": "1"}):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True).stdout.strip().replace("\n", " | ")
        print(env, out, flush=True)
