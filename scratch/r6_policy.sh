#!/bin/bash
# lists (LDS, three sub-tiles) against lane buffers + sample seeds at BASELINE C2's shape, small k_metrics
OUT=gpurun_out/r6_policy.txt
: > $OUT
for K in ${KS:-10 12 14 16 18 20}; do
  for M in 1000000 1; do
    echo "C2 138493 K=$K lane_min_k=$M" >> $OUT
    RM_DEBUG_LANE_MIN_K=$M NS_K=$K timeout 600 python3 scratch/ns.py C2 138493 4 2>&1 | tail -1 | cut -c1-200 >> $OUT
  done
done
cat $OUT
