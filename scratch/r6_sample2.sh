#!/bin/bash
OUT=gpurun_out/r6_sample2.txt
: > $OUT
for K in 21 32 50 100 256 300 1000; do
  for S in 0 -1; do
    echo "C2 138493 K=$K sample=$S" >> $OUT
    RM_DEBUG_SAMPLE_SEED=$S NS_K=$K timeout 600 python3 scratch/ns.py C2 138493 3 2>&1 | tail -1 | cut -c1-200 >> $OUT
  done
done
for S in 0 -1; do
echo "C4 8192 (K=100) sample=$S" >> $OUT; RM_DEBUG_SAMPLE_SEED=$S timeout 900 python3 scratch/ns.py C4 8192 2 2>&1 | tail -1 | cut -c1-200 >> $OUT
echo "NS 32768 K=100 sample=$S" >> $OUT; RM_DEBUG_SAMPLE_SEED=$S NS_K=100 timeout 900 python3 scratch/ns.py NS 32768 2 2>&1 | tail -1 | cut -c1-200 >> $OUT
echo "C3 65536 sample=$S" >> $OUT; RM_DEBUG_SAMPLE_SEED=$S timeout 900 python3 scratch/ns.py C3 65536 2 2>&1 | tail -1 | cut -c1-200 >> $OUT
echo "C2 K=100 128 factors sample=$S" >> $OUT; RM_DEBUG_SAMPLE_SEED=$S NS_K=100 NS_FACTORS=128 timeout 900 python3 scratch/ns.py C2 138493 2 2>&1 | tail -1 | cut -c1-200 >> $OUT
done
cat $OUT
