#!/bin/bash
# per-kernel average durations of one workload: bash scratch/kstat.sh <workload> <users> [lib]
R=$(pwd); [ -n "$3" ] && export RECOMETRICS_HIP_LIB=$R/$3
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/kstat_$$; rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/scratch/ns.py $1 $2 3 > /dev/null 2>&1
cd $R; python3 -c "
import csv,glob
f=sorted(glob.glob('$D/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:9]: print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
