#!/bin/bash
# rocprofv3 kernel trace of a few replayed steps of any bench task -> per-queue timeline of the last step.  TAG=name BENCH_ARGS="--task seg" bash tools/gpu_timeline2.sh
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl2
timeout -s KILL 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 5 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/tl2.log 2>&1
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/tl2.log | cut -c1-200
python tools/queue_timeline.py gpurun_out/tl2/t_results.db > gpurun_out/${TAG:-tl2}_queues.txt 2>&1
head -60 gpurun_out/${TAG:-tl2}_queues.txt
rm -rf gpurun_out/tl2
