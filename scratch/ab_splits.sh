#!/bin/bash
# sweep time by grid shape: bash scratch/ab_splits.sh <out> <wl> <users> "<S,tail_ublocks,tail_splits> ..."   ("-" = the library's own choice)
out=$1; wl=$2; users=$3; shift; shift; shift
mkdir -p gpurun_out/$out
for round in 1 2; do
for sp in "$@"; do
    if [ "$sp" = "-" ]; then unset RM_DEBUG_SPLITS; else export RM_DEBUG_SPLITS=$sp; fi
    python3 scratch/ns.py $wl $users 4 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('splits=$sp', d['workload'], d['users'], round(d['sweep_ms'],3), round(d['frac'],4), d['tm'])" >> gpurun_out/$out/ab.txt
done
done
unset RM_DEBUG_SPLITS
cat gpurun_out/$out/ab.txt
