#!/bin/bash
# random shapes / options against the oracle under the default and under forced code paths (the switches are read at library load:
# one process per setting): bash scratch/fuzz_campaign.sh [cases per setting]
N=${1:-1000}
run() { echo "== $*"; env "$@" python3 scratch/fuzz.py $N $((RANDOM % 1000 + 100)) 2>&1 | tail -3; }
run RM_NONE=1
run RM_DEBUG_HBM_LISTS=1
run RM_DEBUG_NSUB2=1
run RM_DEBUG_NO_TRAIN_BITS=1
run RM_DEBUG_NO_TEST_MASK=1
run RM_DEBUG_NO_SIDE=1
run RM_DEBUG_EXT_TOPK=1
run RM_DEBUG_RANK_GENERIC=1 RM_DEBUG_NO_FUSED_AUC=1
run RM_DEBUG_NOISE_SEQUENTIAL=1
run RM_BATCH_USERS=1024 RM_DEBUG_ONE_CONTEXT=1
run RM_BATCH_USERS=1024
run RM_BATCH_USERS=1024 RM_DEBUG_NO_NOISE_BESIDE_LAST=1
run RM_STREAM_BUDGET_MB=0
run RM_DEBUG_NO_SPEC=1 RM_DEBUG_NO_PENDING=1
run RM_DEBUG_NO_POS_FLAT=1
run RM_DEBUG_NO_POS_BESIDE=1 RM_DEBUG_NO_EARLY_BITS=1
run FUZZ_TIES=1
