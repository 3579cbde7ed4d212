#!/bin/bash
# instruction-cache counters of the sweep kernel.  usage (GPU box, repo root): bash scratch/pmc_icache.sh <workload> <users> <outdir>
WL=${1:-C2}; USERS=${2:-138493}; OUT=${3:-gpurun_out/pmc_icache_$WL}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/$OUT/p1 -- python3 $R/scratch/ns.py $WL $USERS 1 > $R/$OUT.log 2>&1 || echo failed
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in sorted(glob.glob("%s/p*/*/*counter_collection.csv" % out)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_sweep" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if agg:
        last = sorted(agg, key=int)[-1]
        res.update(agg[last])
print(json.dumps(res, indent=1))
json.dump(res, open("%s/icache.json" % out, "w"), indent=1)
PY
