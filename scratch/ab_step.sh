#!/bin/bash
# whole-step A/B of one library under an environment switch: bash scratch/ab_step.sh <out> "<wl users>;..." <ENVVAR>  (ms per step over 6 steps + kernel parts, three rounds)
out=$1; wls=$2; var=$3
mkdir -p gpurun_out/$out
for round in 1 2 3; do
for mode in off on; do
  IFS=';' read -ra W <<< "$wls"
  for wl in "${W[@]}"; do
    set -- $wl
    if [ $mode = on ]; then export $var=1; else unset $var; fi
    python3 scratch/ns.py $1 $2 6 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$var=$mode', d['workload'], d['users'], 'step_ms', round(d['users']/d['users_per_s']*1e3,3), 'sweep', round(d['sweep_ms'],3), 'prep', round(d['prep_ms'],3), 'fin', round(d['fin_ms'],3))" >> gpurun_out/$out/ab.txt
  done
done
done
unset $var
cat gpurun_out/$out/ab.txt
