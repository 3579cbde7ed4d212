"""C2 at its full size: the host-pointer entry (user batches on two contexts) against the device entry, bit for bit, noise off and on:
python3 scratch/check_host_vs_device.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem, host_problem
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
host = host_problem(m, n, k, mean_c, seed, dtype)
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype, host=host)
trp, tri = host["train"]; tep, tei, tev = host["test"]
want = {name: True for name in binding.METRIC_ORDER}
for noise in (False, True):
    outs = binding.calc_metrics(host["A"], k, host["B"], k, trp, tri, tep, tei, tev, K, want, False, noise, True, 2, 1, 1, 1)
    p.step(binding, torch.cuda.current_stream().cuda_stream, noise=noise)
    torch.cuda.synchronize()
    dev = p.out.cpu().numpy()
    bad = 0
    for i, (name, h) in enumerate(zip(binding.METRIC_ORDER, outs)):
        d = dev[i]
        same = (h.view(np.uint32) == d.view(np.uint32)) | (np.isnan(h) & np.isnan(d))
        bad += int((~same).sum())
    print("noise=%s: %d users x 10 metrics, entries that differ between the host entry and the device entry: %d" % (noise, m, bad))
    assert bad == 0
print("ok")
