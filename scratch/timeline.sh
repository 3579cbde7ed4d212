#!/bin/bash
# kernel timeline of ONE step (the last one) of a workload: bash scratch/timeline.sh <workload> <users> <out.txt>
R=$(pwd); WL=${1:-C2}; USERS=${2:-138493}; OUT=${3:-gpurun_out/timeline_$WL.txt}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/timeline_$$
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/scratch/ns.py $WL $USERS 3 > /dev/null 2>&1
cd $R
python3 - "$D" "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last k_count_long (first kernel of a call) on
starts = [i for i, r in enumerate(rows) if "k_count_long" in r["Kernel_Name"] or "k_classify" in r["Kernel_Name"]]
first = [i for i in starts if "k_count_long" in rows[i]["Kernel_Name"]]
i0 = (first or starts)[-1]
t0 = int(rows[i0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("# start_us  dur_us  queue  kernel   (one step of the workload; times relative to the step's first kernel)\n")
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%9.1f %8.1f  q%s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, str(r.get("Queue_Id", "?")) + "/s" + str(r.get("Stream_Id", "?")), r["Kernel_Name"][:90]))
print(open(sys.argv[2]).read())
PY
