#!/bin/bash
# SQ counters of the sweep kernel (instruction mix and stall buckets); counters only, one pass per group.
# usage (GPU box, repo root): bash scratch/pmc_sq.sh <workload> <users> <outdir>
WL=${1:-C2}; USERS=${2:-138493}; OUT=${3:-gpurun_out/pmc_sq_$WL}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -- python3 $R/scratch/ns.py $WL $USERS 1 > $R/$OUT.p$i.log 2>&1 || echo "pass $i failed"
done
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
        for k_, v_ in agg[last].items(): res.setdefault(k_, v_)      # (the first pass that has a counter names it)
json.dump(res, open("%s/sq.json" % out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
