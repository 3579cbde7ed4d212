#!/bin/bash
# Round 5: profiles PER CONFIG: kernel stats of NS, C3, C4, C5, TUT separately; HBM traffic of the sweep launch at
# C2, NS, C4, C5 (B larger than the Infinity Cache at NS / C4 / C5); SQ counters at C2, C4, C5; timeline of one C2 step.
# usage (GPU box, repo root): bash scratch/r5_profiles.sh [what ...]     what = stats traffic sq timeline (default: all)
O=gpurun_out/r5_profiles; mkdir -p $O
WHAT=${@:-stats traffic sq timeline}
users() { case $1 in C2) echo 138493;; NS) echo 32768;; C3) echo 125000;; C4) echo 8192;; C5) echo 16384;; TUT) echo 10000;; esac; }
for what in $WHAT; do
  case $what in
    stats)
      for w in C2 NS C3 C4 C5 TUT; do
        bash scratch/kstats.sh r5_profiles/k_$w $w $(users $w) 4 > $O/kstats_$w.txt 2>&1
        cp $O/k_$w/kernel_stats.csv $O/r5_kernel_stats_$w.csv 2>/dev/null; cp $O/k_$w/ns.json $O/r5_kernel_stats_$w.run.json 2>/dev/null
      done;;
    traffic)
      for w in C2 NS C4 C5; do
        bash scratch/pmc_traffic.sh $w $(users $w) $O/t_$w > $O/traffic_$w.log 2>&1
        cp $O/t_$w/traffic_$w.json $O/r5_traffic_$w.json 2>/dev/null
      done;;
    sq)
      for w in C2 C4 C5; do
        bash scratch/pmc_sq.sh $w $(users $w) $O/sq_$w > /dev/null 2>&1
        cp $O/sq_$w/sq.json $O/r5_pmc_sq_$w.json 2>/dev/null
      done;;
    timeline)
      bash scratch/timeline.sh C2 138493 $O/tl_C2 > $O/r5_timeline_C2.txt 2>&1;;
  esac
done
rm -rf $O/t_* $O/sq_* $O/k_*/p* $O/tl_C2 2>/dev/null
ls -la $O
