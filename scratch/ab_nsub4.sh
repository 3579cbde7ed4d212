#!/bin/bash
# A/B of the four-sub-tile block (RM_DEBUG_NSUB4) against the default three, interleaved rounds on one box: bash scratch/ab_nsub4.sh [rounds] [workload users]
R=${1:-6}; WL=${2:-C2}; U=${3:-138493}
for i in $(seq 1 $R); do
  for v in 0 1; do
    if [ $v = 1 ]; then export RM_DEBUG_NSUB4=1; else unset RM_DEBUG_NSUB4; fi
    python3 scratch/ns.py $WL $U 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('nsub4=$v', 'sweep %.3f ms' % d['sweep_ms'], 'frac %.4f' % d['frac'], 'prep %.3f fin %.3f' % (d['prep_ms'], d['fin_ms']), 'blocks', d['tm']['sweep_blocks'], 'lds', d['tm']['lds_bytes'], 'splits', d['tm']['item_splits'])"
  done
done
