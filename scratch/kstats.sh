#!/bin/bash
# kernel stats of one workload: bash scratch/kstats.sh <out> <wl> <users> [steps]
O=gpurun_out/$1; mkdir -p $O; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o p -- python3 $R/scratch/ns.py $2 $3 ${4:-6} > $R/$O/ns.json 2> $R/$O/prof.err
find /tmp/prof_k -name "*kernel_stats.csv" -exec cp {} $R/$O/kernel_stats.csv \;
cd $R
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(4), ("%.1f" % (float(r['AverageNs']) / 1e3)).rjust(9), "us", r['Percentage'])
PY
