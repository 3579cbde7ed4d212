run() { echo "== $*"; env "$@" python3 scratch/fuzz.py 500 $((RANDOM % 1000 + 100)) 2>&1 | tail -2; }
run RM_NONE=1
run RM_DEBUG_NO_SIDE=1
run RM_DEBUG_NO_POS_BESIDE=1
run RM_BATCH_USERS=1024
run RM_DEBUG_NO_POS_FLAT=1 RM_DEBUG_NO_SIDE=1
run RM_DEBUG_NO_EARLY_BITS=1
run FUZZ_TIES=1
