#!/bin/bash
# A/B of the sample seeds (RM_DEBUG_SAMPLE_SEED = items of the sample; 0 = none) at BASELINE C2's shape
OUT=gpurun_out/r6_sample.txt
: > $OUT
timeout 900 python3 -m pytest tests -m gpu -x -q -k "lane or k_metrics or random_problem or golden" 2>&1 | tail -3 >> $OUT
for K in ${KS:-32 100 256}; do
  for S in ${SS:-0 1024 2048 4096}; do
    echo "C2 138493 K=$K sample=$S" >> $OUT
    RM_DEBUG_SAMPLE_SEED=$S NS_K=$K timeout 600 python3 scratch/ns.py C2 138493 3 2>&1 | tail -1 | cut -c1-200 >> $OUT
  done
done
cat $OUT
