#!/bin/bash
# refresh of the default-workload profiles after late kernel changes (GPU box, repo root)
O=gpurun_out/r2_profiles; mkdir -p $O
R=$(pwd)
python bench.py > $O/r2_bench_default.json 2> $O/bench_default.err
python bench.py --workload NS --no-extra > $O/r2_bench_NS.json 2> $O/bench_NS.err
python bench.py --workload C3 --no-extra > $O/r2_bench_C3.json 2> $O/bench_C3.err
bash scratch/profile_stats.sh   # kernel stats of the timed launches only (gpurun_out/r2_stats)
bash scratch/pmc_traffic.sh C2 138493 $O/traffic_C2 > $O/traffic_C2.log 2>&1
bash scratch/pmc_sq.sh C2 138493 $O/sq_C2 > /dev/null 2>&1
bash scratch/pmc_sq.sh NS 32768 $O/sq_NS > /dev/null 2>&1
bash scratch/pmc_sq.sh C4 8192 $O/sq_C4 > /dev/null 2>&1
bash scratch/pmc_sq.sh C5 16384 $O/sq_C5 > /dev/null 2>&1
cp $O/traffic_C2/traffic_C2.json $O/r2_traffic_C2.json 2>/dev/null
for w in C2 NS C4 C5; do cp $O/sq_$w/sq.json $O/r2_pmc_sq_$w.json 2>/dev/null; done
rm -rf $O/traffic_C2 $O/sq_C2 $O/sq_NS $O/sq_C4 $O/sq_C5 $O/*.p1.log $O/*.p2.log $O/*.p3.log
python bench.py --workload C4 --users 16384 --no-extra > $O/r2_bench_C4.json 2> $O/bench_C4.err
python bench.py --workload C5 --users 50000 --no-extra > $O/r2_bench_C5.json 2> $O/bench_C5.err
bash scratch/pmc_traffic.sh NS 32768 $O/traffic_NS > $O/traffic_NS.log 2>&1
cp $O/traffic_NS/traffic_NS.json $O/r2_traffic_NS.json 2>/dev/null; rm -rf $O/traffic_NS
