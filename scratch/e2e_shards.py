"""host-pointer call (PCIe inclusive) at C2 with 1, 2, 3 virtual shards on device 0"""
import sys, os, json, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem, e2e_host_measure
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
m, n, k, dtype, K, mean_c, seed = CONFIGS[wl]
if len(sys.argv) > 2: m = int(sys.argv[2])
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype)
for devs in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0], [0]):
    binding.set_devices(devs)
    r = e2e_host_measure(binding, p.host, p.k, p.K, p.dtype, reps=5)
    print(len(devs), "shards", round(r["first_call_ms"], 2), round(r["steady_ms"], 2))
