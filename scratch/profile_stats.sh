#!/bin/bash
# kernel stats of the timed launches only (no north-star / noise-on extra legs): GPU box, repo root
O=gpurun_out/${1:-r3}_stats; RND=${1:-r3}; mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in C2 NS C3 C5; do
  extra=""; [ $w = C3 ] && extra="--users 125000"; [ $w = C5 ] && extra="--users 50000"
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o p -- python3 $R/bench.py --workload $w $extra --no-cpu --no-extra --no-e2e > $R/$O/bench_$w.json 2> $R/$O/prof_$w.err
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $R/$O/${RND}_bench_${w}_kernel_stats.csv \;
done
