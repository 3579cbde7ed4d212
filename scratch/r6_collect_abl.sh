#!/bin/bash
# what k_collect_topk's parts cost (timing builds: wrong results): bash scratch/r6_collect_abl.sh "K list" "lib list"
for K in ${1:-100 256}; do
for L in "" ${2:-gonly nogather}; do
  echo "== K=$K lib=$L"
  if [ -n "$L" ]; then export RECOMETRICS_HIP_LIB=$PWD/scratch/libs/lib_abl_$L.so; else unset RECOMETRICS_HIP_LIB; fi
  NS_K=$K bash scratch/kstats.sh ks_abl C2 138493 3 | grep -E "collect|k_finalize<" 
done
done
