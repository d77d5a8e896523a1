#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl3
timeout -s KILL 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl3 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 5 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/tl3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/tail_timeline.py gpurun_out/tl3/t_results.db ${WIN:-1.5} > gpurun_out/${TAG:-r05_u_tail}.txt 2>&1
rm -rf gpurun_out/tl3
cat gpurun_out/${TAG:-r05_u_tail}.txt
