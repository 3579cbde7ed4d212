#!/bin/bash
# A/B of libraries: bash scratch/ab_libs.sh <out> "<wl users>;..." <rounds> <lib1> <lib2> ...   (min sweep ms over 3 steps)
out=$1; wls=$2; rounds=$3; shift; shift; shift
libs=("$@")
mkdir -p gpurun_out/$out
for round in $(seq 1 $rounds); do
for lib in "${libs[@]}"; do
  IFS=';' read -ra W <<< "$wls"
  for wl in "${W[@]}"; do
    set -- $wl
    RECOMETRICS_HIP_LIB=$PWD/$lib python3 scratch/ns.py $1 $2 3 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['workload'], d['users'], round(d['sweep_ms'],3), round(d['frac'],4), round(d['users_per_s']))" >> gpurun_out/$out/ab.txt
  done
done
done
cat gpurun_out/$out/ab.txt
