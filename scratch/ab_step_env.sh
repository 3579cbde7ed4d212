#!/bin/bash
# whole-step A/B of environment switches on one library: bash scratch/ab_step_env.sh <out> "<wl users>;..." <rounds> "<ENV=1 ...>" "<...>"   ("-" = no switch)
out=$1; wls=$2; rounds=$3; shift; shift; shift
variants=("$@")
mkdir -p gpurun_out/$out
for round in $(seq 1 $rounds); do
for envs in "${variants[@]}"; do
  IFS=';' read -ra W <<< "$wls"
  for wl in "${W[@]}"; do
    read -r name users <<< "$wl"
    e=""; [ "$envs" != "-" ] && e="$envs"
    env $e python3 scratch/ns.py $name $users 8 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$envs]', d['workload'], d['users'], 'step_ms', round(d['users']/d['users_per_s']*1e3,3), 'sweep', round(d['sweep_ms'],3), 'prep', round(d['prep_ms'],3), 'fin', round(d['fin_ms'],3))" >> gpurun_out/$out/ab.txt
  done
done
done
cat gpurun_out/$out/ab.txt
