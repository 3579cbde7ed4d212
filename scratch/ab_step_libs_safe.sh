#!/bin/bash
# whole-step A/B of libraries with timeouts: bash scratch/ab_step_libs_safe.sh <out> "<wl users>;..." <rounds> <lib1> <lib2> ...
out=$1; wls=$2; rounds=$3; shift; shift; shift
libs=("$@")
mkdir -p gpurun_out/$out
for round in $(seq 1 $rounds); do
for lib in "${libs[@]}"; do
  IFS=';' read -ra W <<< "$wls"
  for wl in "${W[@]}"; do
    read -r name users <<< "$wl"
    RECOMETRICS_HIP_LIB=$PWD/$lib timeout 90 python3 scratch/ns.py $name $users 8 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['workload'], d['users'], 'step_ms', round(d['users']/d['users_per_s']*1e3,3), 'sweep', round(d['sweep_ms'],3), 'prep', round(d['prep_ms'],3), 'fin', round(d['fin_ms'],3))" >> gpurun_out/$out/ab.txt 2>/dev/null || echo "$lib $name FAILED/TIMEOUT" >> gpurun_out/$out/ab.txt
  done
done
done
cat gpurun_out/$out/ab.txt
