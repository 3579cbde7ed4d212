#!/bin/bash
# Round profiles (run on the GPU box from the repo root): bench lines of every BASELINE workload, rocprofv3 kernel stats of the
# default bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters of the sweep.
# Outputs under gpurun_out/r2_profiles/ -- copy what is to be judged into profiles/.
O=gpurun_out/r2_profiles; mkdir -p $O
R=$(pwd)
python bench.py > $O/r2_bench_default.json 2> $O/bench_default.err
python bench.py --workload NS --no-extra > $O/r2_bench_NS.json 2> $O/bench_NS.err
python bench.py --workload C3 --no-extra > $O/r2_bench_C3.json 2> $O/bench_C3.err
python bench.py --workload C5 --users 50000 --no-extra --steps 3 --warmup 1 > $O/r2_bench_C5.json 2> $O/bench_C5.err
python bench.py --workload C4 --no-extra --steps 2 --warmup 1 --cpu-seconds 20 > $O/r2_bench_C4.json 2> $O/bench_C4.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_def -o p -- python3 $R/bench.py --no-cpu > /dev/null 2> $R/$O/prof_default.err; find /tmp/prof_def -name "*kernel_stats.csv" -exec cp {} $R/$O/r2_bench_default_kernel_stats.csv \; )
for w in "NS --no-extra" "C3 --no-extra" "C5 --users 50000 --no-extra --steps 3 --warmup 1"; do set -- $w; wl=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python3 $R/bench.py --workload $wl --no-cpu --no-e2e --parity-users 0 "$@" > /dev/null 2> $R/$O/prof_$wl.err; find /tmp/prof_$wl -name "*kernel_stats.csv" -exec cp {} $R/$O/r2_bench_${wl}_kernel_stats.csv \; )
done
bash scratch/pmc_traffic.sh C2 138493 $O/traffic_C2 > $O/traffic_C2.log 2>&1
bash scratch/pmc_traffic.sh NS 32768 $O/traffic_NS > $O/traffic_NS.log 2>&1
bash scratch/pmc_sq.sh C2 138493 $O/sq_C2 > /dev/null 2>&1
bash scratch/pmc_sq.sh NS 32768 $O/sq_NS > /dev/null 2>&1
cp $O/traffic_C2/traffic_C2.json $O/r2_traffic_C2.json 2>/dev/null
cp $O/traffic_NS/traffic_NS.json $O/r2_traffic_NS.json 2>/dev/null
cp $O/sq_C2/sq.json $O/r2_pmc_sq_C2.json 2>/dev/null
cp $O/sq_NS/sq.json $O/r2_pmc_sq_NS.json 2>/dev/null
rm -rf $O/traffic_C2 $O/traffic_NS $O/sq_C2 $O/sq_NS $O/*.p1.log $O/*.p2.log $O/*.p3.log
ls -la $O
