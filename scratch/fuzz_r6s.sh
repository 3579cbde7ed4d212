#!/bin/bash
# Round 6: the sample seeds of the lane buffers (k_seed_from_sample) forced at every sample size under random shapes and options,
# against the oracle.   bash scratch/fuzz_r6s.sh [cases per setting]
N=${1:-500}
run() { echo "== $*"; env "$@" python3 scratch/fuzz.py $N $((RANDOM % 1000 + 100)) 2>&1 | tail -3; }
run RM_DEBUG_SAMPLE_SEED=64 RM_DEBUG_LANE_MIN_K=1
run RM_DEBUG_SAMPLE_SEED=256 RM_DEBUG_LANE_MIN_K=1
run RM_DEBUG_SAMPLE_SEED=64 RM_DEBUG_LANE_MIN_K=1 RM_DEBUG_LANE_CAP_MIN=1 FUZZ_TIES=1
run RM_DEBUG_SAMPLE_SEED=256 RM_DEBUG_LANE_MIN_K=1 RM_DEBUG_NO_TRAIN_BITS=1 RM_DEBUG_NO_SEED=1
run RM_DEBUG_SAMPLE_SEED=64 RM_DEBUG_LANE_CAP_MIN=1 RM_DEBUG_SPLITS=3,2,5
run RM_DEBUG_SAMPLE_SEED=1024
run RM_DEBUG_SAMPLE_SEED=256 RM_BATCH_USERS=1024 RM_DEBUG_LANE_MIN_K=1
