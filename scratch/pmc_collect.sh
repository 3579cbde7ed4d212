#!/bin/bash
# counters of k_collect_topk (one launch of C2's shape at K = $NS_K): bash scratch/pmc_collect.sh <out dir> [lib]
OUT=${1:-gpurun_out/pmc_collect}; R=$(pwd); mkdir -p $R/$OUT
[ -n "$2" ] && export RECOMETRICS_HIP_LIB=$R/scratch/libs/lib_abl_$2.so
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -- python3 $R/scratch/ns.py C2 138493 1 > $R/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_collect_topk" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
disp = max(1, min(n.values()) if n else 1)
for k in sorted(tot): print("%-40s %16.0f  (over %d dispatches)" % (k, tot[k], n[k]))
PY
