#!/bin/bash
# A/B of one library under an environment switch: bash scratch/ab_env.sh <out> "<wl users>;..." <ENVVAR>   (min sweep ms over 4 steps, two rounds)
out=$1; wls=$2; var=$3
mkdir -p gpurun_out/$out
for round in 1 2; do
for mode in off on; do
  IFS=';' read -ra W <<< "$wls"
  for wl in "${W[@]}"; do
    set -- $wl
    if [ $mode = on ]; then export $var=1; else unset $var; fi
    python3 scratch/ns.py $1 $2 4 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$var=$mode', d['workload'], d['users'], round(d['sweep_ms'],3), round(d['frac'],4), round(d['users_per_s']))" >> gpurun_out/$out/ab.txt
  done
done
done
unset $var
cat gpurun_out/$out/ab.txt
