#!/bin/bash
# bash scratch/ab3.sh <out>: doubling ablations of the top-K path, old scan (RM_DEBUG_NO_ROW_DUMP=1) -- timing only
out=$1; mkdir -p gpurun_out/$out
export RM_DEBUG_NO_ROW_DUMP=1
for round in 1 2; do
for lib in recometrics_amd/csrc/librecometrics_hip.so scratch/libs/lib_abl_scan2.so scratch/libs/lib_abl_merge2.so; do
    RECOMETRICS_HIP_LIB=$PWD/$lib python3 scratch/ns.py C2 138493 4 2>>gpurun_out/$out/err.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['workload'], d['users'], round(d['sweep_ms'],3), round(d['frac'],4), round(d['users_per_s']))" >> gpurun_out/$out/ab.txt
done
done
RM_PRINT_STATS=1 RM_STATS_FN=rm_debug_stats_n3 RECOMETRICS_HIP_LIB=$PWD/scratch/libs/lib_abl_stats.so python3 scratch/ns.py C2 138493 1 2>>gpurun_out/$out/err.txt | tail -1 >> gpurun_out/$out/ab.txt
unset RM_DEBUG_NO_ROW_DUMP
RM_PRINT_STATS=1 RM_STATS_FN=rm_debug_stats_n3 RECOMETRICS_HIP_LIB=$PWD/scratch/libs/lib_abl_stats.so python3 scratch/ns.py C2 138493 1 2>>gpurun_out/$out/err.txt | tail -1 >> gpurun_out/$out/ab.txt
cat gpurun_out/$out/ab.txt
