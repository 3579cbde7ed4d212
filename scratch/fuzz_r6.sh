#!/bin/bash
# Round 6: the lane buffers / lane-parallel selections / k_collect_topk under random shapes and options, against the oracle (metrics
# bitwise, ordered top-K lists, positive ranks): default policy, every k_metrics through the lane buffers, the smallest buffers that
# work (a selection every few tiles), exact ties, forced item splits, host batches.   bash scratch/fuzz_r6.sh [cases per setting]
N=${1:-500}
run() { echo "== $*"; env "$@" python3 scratch/fuzz.py $N $((RANDOM % 1000 + 100)) 2>&1 | tail -3; }
run RM_NONE=1
run RM_DEBUG_LANE_MIN_K=1
run RM_DEBUG_LANE_CAP_MIN=1
run RM_DEBUG_LANE_MIN_K=1 RM_DEBUG_LANE_CAP_MIN=1
run RM_DEBUG_LANE_MIN_K=1 RM_DEBUG_LANE_CAP_MIN=1 FUZZ_TIES=1
run RM_DEBUG_LANE_CAP_MIN=1 RM_DEBUG_SPLITS=3,2,5
run RM_DEBUG_LANE_MIN_K=1 RM_DEBUG_NO_TRAIN_BITS=1 RM_DEBUG_NO_SEED=1
run RM_DEBUG_LANE_MIN_K=1000000
run RM_BATCH_USERS=1024 RM_DEBUG_LANE_CAP_MIN=1
run RM_DEBUG_EXT_TOPK=1
