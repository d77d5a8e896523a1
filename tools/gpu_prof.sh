#!/bin/bash
# bench (no CPU baseline) + rocprof kernel stats; outputs under gpurun_out/
# the traced run keeps ONE stream (SIDE=0) so that per-kernel durations are not inflated by the overlapped weight-gradient stream
mkdir -p gpurun_out
python bench.py --steps ${STEPS:-10} --warmup 3 --cpu-budget 0 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
CINEMA_SIDE_WGRAD=${SIDE:-0} rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o mae -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/prof/mae_results.db 5 > gpurun_out/prof_summary.txt
head -40 gpurun_out/prof_summary.txt
