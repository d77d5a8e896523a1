#!/bin/bash
# WINDOWS="A1:B1 A2:B2" TAG=name bash tools/gpu_window.sh
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl4
timeout -s KILL 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl4 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 5 --cpu-budget 0 --profile-steps 0 --prewarm 0 --no-secondary $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/tl4.log 2>&1
cd $GRAFT_REPO_ROOT
: > gpurun_out/${TAG:-r05_v_windows}.txt
for w in $WINDOWS; do python tools/window_timeline.py gpurun_out/tl4/t_results.db "${w%%:*}" "${w##*:}" >> gpurun_out/${TAG:-r05_v_windows}.txt 2>&1; done
rm -rf gpurun_out/tl4
cat gpurun_out/${TAG:-r05_v_windows}.txt
