#!/bin/bash
# per-kernel time of one bench configuration: scratch/kstat2.sh <outdir> <bench args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-extra --parity-users 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$out/err.log
find /tmp/prof_$out -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/$out/kernel_stats.csv \;
