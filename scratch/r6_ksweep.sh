#!/bin/bash
# K sweep at BASELINE C2's shape (device-resident, all ten metrics) + the configs whose K is beyond the LDS lists:
#   bash scratch/r6_ksweep.sh [out file] ["K list"]
OUT=${1:-gpurun_out/r6_ksweep.txt}
KS=${2:-"10 20 32 33 50 100 256 300 1000"}
: > $OUT
for K in $KS; do
  echo "C2 138493 K=$K" >> $OUT
  NS_K=$K timeout 600 python3 scratch/ns.py C2 138493 3 2>&1 | tail -1 | cut -c1-330 >> $OUT
done
if [ -z "$R6_C2_ONLY" ]; then
echo "C4 8192 (K=100)" >> $OUT; timeout 900 python3 scratch/ns.py C4 8192 2 2>&1 | tail -1 | cut -c1-330 >> $OUT
echo "C5 16384 (K=50)" >> $OUT; timeout 900 python3 scratch/ns.py C5 16384 2 2>&1 | tail -1 | cut -c1-330 >> $OUT
echo "NS 32768 K=100" >> $OUT; NS_K=100 timeout 900 python3 scratch/ns.py NS 32768 2 2>&1 | tail -1 | cut -c1-330 >> $OUT
fi
cat $OUT
