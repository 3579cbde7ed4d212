#!/bin/bash
# kernel stats of the noise-on step at C2: bash scratch/kstats_noise.sh <out>
O=gpurun_out/$1; mkdir -p $O; R=$(pwd)
cat > /tmp/noise_only.py <<PY
import sys, os
sys.path.insert(0, "$R")
import torch
from recometrics_amd import _binding as binding
from recometrics_amd.synth import CONFIGS
from bench import DeviceProblem
m, n, k, dtype, K, mean_c, seed = CONFIGS["C2"]
torch.cuda.set_device(0); binding.load(); binding.set_device(0)
p = DeviceProblem(torch, torch.device("cuda", 0), m, n, k, mean_c, seed, K, dtype)
st = torch.cuda.current_stream().cuda_stream
for _ in range(7): p.step(binding, st, noise=True)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o p -- python3 /tmp/noise_only.py > $R/$O/out.txt 2> $R/$O/prof.err
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $R/$O/kernel_stats.csv \;
cd $R
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step", tot / 7 / 1e6)
for r in rows[:40]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(4), ("%.1f" % (float(r['AverageNs']) / 1e3)).rjust(9), "us", ("%.3f" % (float(r['TotalDurationNs']) / 7 / 1e6)).rjust(7), "ms/step")
PY
