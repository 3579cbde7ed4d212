#!/bin/bash
# Round 6: profiles PER CONFIG (as round 5) + the requests without ROC / PR-AUC + the K sweep at C2's shape: kernel stats of NS, C3, C4, C5, TUT separately; HBM traffic of the sweep launch at
# C2, NS, C4, C5 (B larger than the Infinity Cache at NS / C4 / C5); SQ counters at C2, C4, C5; timeline of one C2 step.
# usage (GPU box, repo root): bash scratch/r6_profiles.sh [what ...]     what = stats traffic sq timeline (default: all)
O=gpurun_out/r6_profiles; mkdir -p $O
WHAT=${@:-stats ksweep traffic sq timeline}
users() { case $1 in C2) echo 138493;; NS) echo 32768;; C3) echo 125000;; C4) echo 8192;; C5) echo 16384;; TUT) echo 10000;; esac; }
for what in $WHAT; do
  case $what in
    stats)
      for w in C2 NS C3 C4 C5 TUT; do
        bash scratch/kstats.sh r6_profiles/k_$w $w $(users $w) 4 > $O/kstats_$w.txt 2>&1
        cp $O/k_$w/kernel_stats.csv $O/r6_kernel_stats_$w.csv 2>/dev/null; cp $O/k_$w/ns.json $O/r6_kernel_stats_$w.run.json 2>/dev/null
      done
      # the k_sweep<..., AUC = false, ...> family: the API's default request (precision, average_precision, ndcg; noise on) at C2's and the
      # tutorial's shape, the eight top-K metrics at the north-star shape (NS_DROP = indices into METRIC_ORDER that are NOT asked for)
      NS_DROP=1,2,4,6,7,8,9 NS_NOISE=1 bash scratch/kstats.sh r6_profiles/k_C2_noauc C2 138493 4 > $O/kstats_C2_noauc.txt 2>&1
      cp $O/k_C2_noauc/kernel_stats.csv $O/r6_kernel_stats_C2_defaults_noauc.csv 2>/dev/null; cp $O/k_C2_noauc/ns.json $O/r6_kernel_stats_C2_defaults_noauc.run.json 2>/dev/null
      NS_DROP=1,2,4,6,7,8,9 NS_NOISE=1 bash scratch/kstats.sh r6_profiles/k_TUT_noauc TUT 10000 4 > $O/kstats_TUT_noauc.txt 2>&1
      cp $O/k_TUT_noauc/kernel_stats.csv $O/r6_kernel_stats_TUT_defaults_noauc.csv 2>/dev/null; cp $O/k_TUT_noauc/ns.json $O/r6_kernel_stats_TUT_defaults_noauc.run.json 2>/dev/null
      NS_DROP=8,9 bash scratch/kstats.sh r6_profiles/k_NS_noauc NS 32768 4 > $O/kstats_NS_noauc.txt 2>&1
      cp $O/k_NS_noauc/kernel_stats.csv $O/r6_kernel_stats_NS_noauc.csv 2>/dev/null; cp $O/k_NS_noauc/ns.json $O/r6_kernel_stats_NS_noauc.run.json 2>/dev/null
      # C2's shape at K = 100 (lane buffers + k_collect_topk)
      NS_K=100 bash scratch/kstats.sh r6_profiles/k_C2_K100 C2 138493 4 > $O/kstats_C2_K100.txt 2>&1
      cp $O/k_C2_K100/kernel_stats.csv $O/r6_kernel_stats_C2_K100.csv 2>/dev/null; cp $O/k_C2_K100/ns.json $O/r6_kernel_stats_C2_K100.run.json 2>/dev/null;;
    ksweep)
      bash scratch/r6_ksweep.sh $O/r6_ksweep_C2.txt "10 20 21 32 50 100 256 300 1000" > /dev/null 2>&1;;
    traffic)
      for w in C2 NS C4 C5; do
        bash scratch/pmc_traffic.sh $w $(users $w) $O/t_$w > $O/traffic_$w.log 2>&1
        cp $O/t_$w/traffic_$w.json $O/r6_traffic_$w.json 2>/dev/null
      done;;
    sq)
      for w in C2 C4 C5; do
        bash scratch/pmc_sq.sh $w $(users $w) $O/sq_$w > /dev/null 2>&1
        cp $O/sq_$w/sq.json $O/r6_pmc_sq_$w.json 2>/dev/null
      done;;
    timeline)
      bash scratch/timeline.sh C2 138493 $O/tl_C2 > $O/r6_timeline_C2.txt 2>&1;;
  esac
done
rm -rf $O/t_* $O/sq_* $O/k_*/p* $O/tl_C2 2>/dev/null
ls -la $O
