#!/bin/bash
# rocprof kernel stats of the config-4 (ConvUNetR fine-tuning) step, one stream; output gpurun_out/seg_kernel_stats.txt
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/profseg
CINEMA_SIDE_WGRAD=${SIDE:-0} rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/profseg -o seg -- python $GRAFT_REPO_ROOT/bench.py --task seg --steps 3 --warmup 2 --cpu-budget 0 --profile-steps 0 --prewarm 0 > $GRAFT_REPO_ROOT/gpurun_out/profseg.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/profseg/seg_results.db 5 > gpurun_out/seg_kernel_stats.txt
rm -rf gpurun_out/profseg
head -30 gpurun_out/seg_kernel_stats.txt
