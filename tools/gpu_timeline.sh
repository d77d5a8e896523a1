#!/bin/bash
# rocprofv3 kernel trace of a few recorded (replayed) steps -> per-phase timeline of the last step
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl -o mae -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --cpu-budget 0 --profile-steps 0 --prewarm 0 $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/tl.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/tl.log | cut -c1-200
python tools/phase_timeline.py gpurun_out/tl/mae_results.db
ls -la gpurun_out/tl
