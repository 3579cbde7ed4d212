#!/bin/bash
# the tutorial shape (10,000 x 160,112 x 50): item-range splits by hand against the cost model's choice
OUT=gpurun_out/r6_tut_splits.txt
: > $OUT
for SP in "" 3 6 16 "5,28,8" "5,28,9" "6,37,8" "11,9,16" "3,0,0" "15,0,0" "16,0,0" "6,36,7" "2,0,0" "1,0,0"; do
  echo "TUT splits=$SP" >> $OUT
  if [ -n "$SP" ]; then export RM_DEBUG_SPLITS=$SP; else unset RM_DEBUG_SPLITS; fi
  timeout 600 python3 scratch/ns.py TUT 10000 6 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['sweep_ms'],3), round(d['frac'],4), round(d['users_per_s']), d['tm']['item_splits'], d['tm']['sweep_blocks'])" >> $OUT
done
cat $OUT
