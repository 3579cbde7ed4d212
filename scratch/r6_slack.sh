#!/bin/bash
OUT=gpurun_out/r6_slack.txt
: > $OUT
run() { # wl users steps
  for SL in 20 10 5 2; do
    echo "$1 $2 slack=$SL $3" >> $OUT
    RM_DEBUG_SPLIT_SLACK=$SL timeout 900 python3 scratch/ns.py $1 $2 $4 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['sweep_ms'],3), round(d['frac'],4), round(d['users_per_s']), d['tm']['item_splits'], d['tm']['sweep_blocks'])" >> $OUT
  done
}
run TUT 10000 "" 6
run C2 138493 "" 5
run NS 32768 "" 3
run C3 125000 "" 2
run C4 8192 "" 2
run C5 16384 "" 2
NS_K=100 run C2 138493 K100 4
run C1 1000 "" 6
run C2 20000 "" 6
run C2 60000 "" 6
cat $OUT
