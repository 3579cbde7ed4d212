#!/bin/bash
# A/B timing of library variants: bash scratch/ab.sh <out> <lib1> <lib2> ...   (each: C2 full + NS 8192 users, 5 reps, min sweep ms)
out=$1; shift
mkdir -p gpurun_out/$out
for lib in "$@"; do
  for wl in "C2 138493" "NS 16384"; do
    set -- $wl
    RECOMETRICS_HIP_LIB=$PWD/$lib python3 scratch/ns.py $1 $2 5 2>&1 | tail -1 | sed "s|^|$lib $1 |" >> gpurun_out/$out/ab.txt
  done
done
cat gpurun_out/$out/ab.txt
