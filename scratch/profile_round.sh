#!/bin/bash
# end-of-round measurements (GPU box, repo root): bash scratch/profile_round.sh r3
RND=${1:-r3}; O=gpurun_out/${RND}_round; mkdir -p $O
python bench.py > $O/${RND}_bench_default.json 2> $O/bench_default.err
for w in NS C3 C4 C5; do
  extra=""; [ $w = C3 ] && extra="--users 125000"; [ $w = C4 ] && extra="--users 16384"; [ $w = C5 ] && extra="--users 50000"
  python bench.py --workload $w $extra --no-extra --no-e2e --cpu-seconds 6 > $O/${RND}_bench_$w.json 2> $O/bench_$w.err
done
bash scratch/profile_stats.sh $RND > /dev/null 2>&1
cp gpurun_out/${RND}_stats/${RND}_bench_*_kernel_stats.csv $O/ 2>/dev/null
bash scratch/pmc_traffic.sh C2 138493 $O/traffic_C2 > /dev/null 2>&1; cp $O/traffic_C2/traffic_C2.json $O/${RND}_traffic_C2.json
bash scratch/pmc_traffic.sh NS 32768 $O/traffic_NS > /dev/null 2>&1; cp $O/traffic_NS/traffic_NS.json $O/${RND}_traffic_NS.json
bash scratch/pmc_sq.sh C2 138493 $O/sq_C2 > /dev/null 2>&1; cp $O/sq_C2/sq.json $O/${RND}_pmc_sq_C2.json
bash scratch/pmc_sq.sh C5 16384 $O/sq_C5 > /dev/null 2>&1; cp $O/sq_C5/sq.json $O/${RND}_pmc_sq_C5.json
rm -rf $O/traffic_C2 $O/traffic_NS $O/sq_C2/p* $O/sq_C5/p* gpurun_out/${RND}_stats
ls $O
